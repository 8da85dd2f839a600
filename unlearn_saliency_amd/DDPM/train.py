"""`python train.py --config cifar10_saliency_unlearn.yml --ckpt_folder F --label_to_forget 0
--mode generate_mask|saliency_unlearn [--mask_path M --alpha 1e-3 --method rl]`

Same flags as the reference's DDPM/train.py:15-93.  Modes on the SalUn hot path are implemented
(generate_mask, saliency_unlearn); train / forget / retrain are pre-training and EWC baselines
 and exit with a scope note.
Multi-GPU: launch with torchrun (one process per GPU); the reference's nn.DataParallel is not used."""
import argparse
import logging
import os
import sys
import traceback

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import unlearn_saliency_amd.DDPM  # noqa: F401
    __package__ = "unlearn_saliency_amd.DDPM"

import numpy as np
import torch

from .. import dist as sdist
from .functions import get_config_and_setup_dirs, get_mask_config_and_setup_dirs
from .runners.diffusion import Diffusion

_HERE = os.path.dirname(os.path.abspath(__file__))

_FLAGS = [
    ("--config", dict(type=str, required=True, help="Path to the config file (relative to configs/)")),
    ("--ckpt_folder", dict(type=str, help="Path to folder with pretrained model. Only for forgetting training.")),
    ("--mode", dict(type=str, default="train", help="train | forget | retrain | saliency_unlearn | generate_mask")),
    ("--label_to_forget", dict(type=int, default=0, help="Class label 0-9 to forget.")),
    ("--seed", dict(type=int, default=1234, help="Random seed")),
    ("--verbose", dict(type=str, default="info", help="Verbose level: info | debug | warning | critical")),
    ("--sample_type", dict(type=str, default="generalized")),
    ("--skip_type", dict(type=str, default="uniform")),
    ("--timesteps", dict(type=int, default=1000)),
    ("--eta", dict(type=float, default=1.0)),
    ("--cond_scale", dict(type=float, default=2.0, help="classifier-free guidance conditional strength")),
    ("--sequence", dict(action="store_true")),
    ("--alpha", dict(type=float, default=1.0, help="forget loss vs remain loss")),
    ("--mask_path", dict(type=str, default=None, help="the path to store mask")),
    ("--method", dict(type=str, default=None, help="the method to unlearn (rl | ga)")),
    ("--uc", dict(type=bool, default=True)),
    ("--negative_guidance", dict(type=float, default=7.5)),
    ("--mask_ratio", dict(type=float, default=0.5)),
    ("--sparse", dict(type=bool, default=False)),
    # build extensions
    ("--synthetic", dict(action="store_true", help="random-init U-Net + synthetic CIFAR-shaped data (benchmarks)")),
    ("--n_iters", dict(type=int, default=None, help="override config.training.n_iters")),
    ("--library_conv", dict(action="store_true", help="use the library (MIOpen) convolutions instead of the MFMA kernels")),
]


def parse_args_and_config(argv=None):
    parser = argparse.ArgumentParser(description=__doc__)
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    args = parser.parse_args(argv)
    cfg = args.config if os.path.exists(args.config) else os.path.join(_HERE, "configs", args.config)
    if args.mode == "saliency_unlearn":
        config = get_mask_config_and_setup_dirs(args, cfg)
    else:
        config = get_config_and_setup_dirs(cfg)
    if args.n_iters is not None:
        config.training.n_iters = args.n_iters
    level = getattr(logging, args.verbose.upper(), None)
    if not isinstance(level, int):
        raise ValueError("level {} not supported".format(args.verbose))
    fmt = logging.Formatter("%(levelname)s - %(filename)s - %(asctime)s - %(message)s")
    logger = logging.getLogger()
    for h in (logging.StreamHandler(), logging.FileHandler(os.path.join(config.log_dir, "stdout.txt"))):
        h.setFormatter(fmt)
        logger.addHandler(h)
    logger.setLevel(level)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.seed)
    torch.backends.cudnn.benchmark = True
    return args, config


def main(argv=None):
    sdist.init_from_env()
    args, config = parse_args_and_config(argv)
    try:
        runner = Diffusion(args, config)
        if args.mode == "saliency_unlearn":
            runner.saliency_unlearn()
        elif args.mode == "generate_mask":
            runner.generate_mask()
        elif args.mode == "forget":  # EWC / Selective Amnesia on the fused penalty kernel
            runner.train_forget()
        else:
            raise NotImplementedError(f"--mode {args.mode}: pre-training / retrain are outside the SalUn hot "
                                      "path of this build (SURVEY.md §2 D1)")
    except Exception:
        logging.error(traceback.format_exc())
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
