"""`python fim.py --config cifar10_fim.yml --ckpt_folder F --n_chunks 20` — diagonal Fisher information of the
ε-MSE objective (reference DDPM/fim.py:14-95 -> Diffusion.save_fim).  Output: {ckpt_folder}/fisher_dict.pkl."""
import argparse
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import unlearn_saliency_amd.DDPM  # noqa: F401
    __package__ = "unlearn_saliency_amd.DDPM"

import numpy as np
import torch

from .. import dist as sdist
from .functions import load_config
from .runners.diffusion import Diffusion

_HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__)
    p.add_argument("--config", type=str, required=True)
    p.add_argument("--ckpt_folder", type=str, required=True)
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--n_chunks", type=int, default=20, help="timestep chunks per sample (memory vs speed)")
    p.add_argument("--label_to_forget", type=int, default=0)
    p.add_argument("--synthetic", action="store_true")
    args = p.parse_args(argv)
    sdist.init_from_env()
    cfg = args.config if os.path.exists(args.config) else os.path.join(_HERE, "configs", args.config)
    config = load_config(cfg)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    Diffusion(args, config).save_fim()
    return 0


if __name__ == "__main__":
    sys.exit(main())
