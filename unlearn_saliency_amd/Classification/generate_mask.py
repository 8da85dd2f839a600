"""Phase A — saliency-mask generation for classifiers.

Drop-in for the reference script (Classification/generate_mask.py): same flags
(arg_parser.py), same artefacts — ``{save_dir}/with_{ratio}.pt`` for ratio 0.1 … 1.0,
each a dict ``param_name -> int64 0/1 tensor of the parameter's shape``.

Reference algorithm (generate_mask.py:14-82), per threshold: concatenate |Σ grads|,
two full ``argsort``s of N = 11.17 M, 186 small kernels, one 89 MB ``torch.save``.
Here:  Σ grads lives in one flat fp32 vector (`salun_saliency_accumulate`, one launch per
batch), all ten thresholds come out of ONE radix-select pass set (`salun_mask_topk`,
abs fused, u8 masks), and int64 expansion happens only at the file boundary.
Under data parallel the forget batches are sharded and the accumulator is all-reduced once.
"""
from __future__ import annotations

import os
import sys
import time
from collections import OrderedDict

if __package__ in (None, ""):  # executed as `python generate_mask.py ...` from this directory, like the reference
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import unlearn_saliency_amd.Classification  # noqa: F401
    __package__ = "unlearn_saliency_amd.Classification"

import torch
import torch.nn as nn

from .. import dist as sdist
from .. import ops
from ..flat import arena_of
from . import arg_parser, utils
from .dataset import BatchLoader, split_marked

THRESHOLD_LIST = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]  # generate_mask.py:50


def accumulate_saliency(forget_loader, model, criterion, arena=None) -> torch.Tensor:
    """Flat Σ_batches ∇θ(−criterion(f(x), y)) with the model in eval mode
    (generate_mask.py:25-44).  Each batch contributes the gradient of its *mean* loss; under
    data parallel a rank's shard contributes (n_local / n_global) of its local mean, and the
    flat accumulator is summed across ranks once at the end."""
    arena = arena or arena_of(model)
    dev = arena.device
    acc = arena.new_like()
    model.eval()
    ws = sdist.world_size()
    for image, target in forget_loader:
        image, target = image.to(dev, non_blocking=True), target.to(dev, non_blocking=True)
        n_local = image.size(0)
        scale = 1.0
        if ws > 1:
            sh = getattr(forget_loader, "last_shard", None)
            if sh is not None:  # the loader knows the global batch size: no collective, no host sync
                scale = n_local / float(sh[2])
            else:
                cnt = torch.tensor([float(n_local)], device=dev)
                sdist.all_reduce_sum_(cnt)
                scale = n_local / float(cnt.item())
        if n_local == 0:
            continue
        loss = -criterion(model(image), target)
        arena.zero_grad()
        loss.backward()
        ops.saliency_accumulate(acc, arena.grads, scale)
    sdist.all_reduce_sum_(acc)
    return acc


def masks_from_saliency(acc: torch.Tensor, ratios) -> "OrderedDict[float, torch.Tensor]":
    """ratio -> flat u8 mask; k = int(N * ratio) in Python double arithmetic (generate_mask.py:60)."""
    n = acc.numel()
    ratios = list(ratios)
    out: "OrderedDict[float, torch.Tensor]" = OrderedDict()
    for s in range(0, len(ratios), 16):  # SALUN_MAX_THRESHOLDS per call
        chunk = ratios[s:s + 16]
        masks = ops.mask_topk(acc, [int(n * r) for r in chunk], check=True)  # raises instead of returning garbage
        out.update(zip(chunk, masks))
    return out


def save_gradient_ratio(data_loaders, model, criterion, args):
    """Same name / signature / outputs as the reference function (generate_mask.py:14-82)."""
    arena = arena_of(model)
    t0 = time.time()
    acc = accumulate_saliency(data_loaders["forget"], model, criterion, arena)
    ratios = THRESHOLD_LIST
    if getattr(args, "thresholds", None):
        ratios = [float(x) for x in str(args.thresholds).split(",")]
    flat_masks = masks_from_saliency(acc, ratios)
    torch.cuda.synchronize() if acc.is_cuda else None
    t1 = time.time()
    if sdist.rank() == 0:
        os.makedirs(args.save_dir, exist_ok=True)
        for ratio, m in flat_masks.items():
            hard_dict = arena.unpack_mask(m)  # int64 0/1, parameter shapes, on the compute device like the reference
            torch.save(hard_dict, os.path.join(args.save_dir, "with_{}.pt".format(ratio)))
    sdist.barrier()
    t2 = time.time()
    print(f"mask generation: saliency+top-k {t1 - t0:.3f}s, writing {len(flat_masks)} mask files {t2 - t1:.3f}s")
    return flat_masks


def main(argv=None):
    args = arg_parser.parse_args(argv)
    rk, lrk, ws = sdist.init_from_env()
    if torch.cuda.is_available():
        if ws == 1:
            torch.cuda.set_device(int(args.gpu))
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        raise RuntimeError("generate_mask needs a ROCm device: the saliency kernels have no CPU fallback")
    os.makedirs(args.save_dir, exist_ok=True)
    if args.seed:
        utils.setup_seed(args.seed)
    seed = args.seed
    model, train_loader_full, val_loader, test_loader, marked_loader = utils.setup_model_dataset(args)
    model.to(device)
    if not args.library_conv:
        from ..conv import use_salun_convs
        use_salun_convs(model)  # convolutions on the fp32 MFMA kernels (csrc/salun_conv.hip)
        from ..norm import use_fused_bn
        use_fused_bn(model)  # BN(+add)+ReLU as one node (csrc/salun_norm.hip); SyncBatchNorm layers are left alone

    def replace_loader_dataset(dataset, batch_size=args.batch_size, seed=1, shuffle=True):
        utils.setup_seed(seed)
        return BatchLoader(dataset, batch_size, shuffle, device_resident=bool(args.device_loader), device=device,
                           rank=rk, world_size=ws)

    forget_dataset, retain_dataset = split_marked(marked_loader.dataset)
    forget_loader = replace_loader_dataset(forget_dataset, seed=seed, shuffle=True)
    retain_loader = replace_loader_dataset(retain_dataset, seed=seed, shuffle=True)
    assert len(forget_dataset) + len(retain_dataset) == len(train_loader_full.dataset)
    print(f"number of retain dataset {len(retain_dataset)}")
    print(f"number of forget dataset {len(forget_dataset)}")
    unlearn_data_loaders = OrderedDict(retain=retain_loader, forget=forget_loader, val=val_loader, test=test_loader)

    criterion = nn.CrossEntropyLoss()
    if args.model_path:
        checkpoint = torch.load(args.model_path, map_location=device, weights_only=False)
        if "state_dict" in checkpoint.keys():
            checkpoint = checkpoint["state_dict"]
        model.load_state_dict(checkpoint, strict=False)
    elif not args.synthetic:
        raise ValueError("--model_path is required (the original model's checkpoint)")
    save_gradient_ratio(unlearn_data_loaders, model, criterion, args)


if __name__ == "__main__":
    main()
