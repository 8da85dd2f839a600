"""Shared by the three SD command lines (generate_mask.py, random_label.py, nsfw_removal.py): flag tables copied from
the reference's parsers (names, types, defaults; pinned by tests/golden/cli.json) and the batch sources.

The reference's scripts build their loaders from image folders through the frozen VAE / CLIP encoders
(SD/train-scripts/dataset.py), which are outside the hot-path scope and not available offline; these front-ends take
pre-encoded batches instead:

    --latents FILE    torch.save'd dict {"forget": [(z, c_forget, c_other), ...], "remain": [(z, c), ...]} with
                      z (B,4,64,64) latents and c (B,77,768) CLIP contexts (c_other = the empty-prompt context for
                      mask generation, the pseudo-prompt context for unlearning)
    --synthetic N     N synthetic batches of that shape (benchmarks / smoke runs on a randomly initialised U-Net)
"""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def add_batch_source_flags(parser):
    parser.add_argument("--latents", type=str, default=None, help="file with pre-encoded forget / remain batches")
    parser.add_argument("--synthetic", type=int, default=0, help="use this many synthetic batches instead")
    parser.add_argument("--resident_activations", action="store_true",
                        help="keep every activation in HBM instead of the config's activation checkpointing (use_checkpoint: "
                             "True re-runs each block inside backward): identical results, no recompute; ~50 GB at batch 8")
    parser.add_argument("--bf16", action="store_true", help="bf16 configuration: bf16 NHWC MFMA convolutions, GroupNorm, attention, LayerNorm / GEGLU kernels (K11-K14), fp32 master weights")


def device_of(arg: str) -> str:
    """`--device 4` -> "cuda:4" (the reference does f"cuda:{int(args.device)}")."""
    if not torch.cuda.is_available():
        raise SystemExit("the SD scripts need a ROCm device (no CPU fallback for the measured path)")
    return f"cuda:{int(arg)}"


def batches(args, device, kinds, model=None):
    """-> dict kind -> list of batches on `device`; kinds: {"forget": 3, "remain": 2} tensors per batch.  Synthetic
    batches take their shapes from the model's U-Net configuration (v1: 4 x 64 x 64 latents, 77 x 768 contexts)."""
    if args.latents:
        data = torch.load(args.latents, map_location=device, weights_only=False)
        return {k: [tuple(t.to(device) for t in b) for b in data[k]] for k in kinds}
    if args.synthetic > 0:
        g = torch.Generator(device=device).manual_seed(0)
        mk = lambda *s: torch.randn(*s, device=device, generator=g)
        B = args.batch_size
        unet = model.model.diffusion_model if model is not None else None
        cin = getattr(unet, "in_channels", 4)
        hw = 64 if unet is None or unet.model_channels == 320 else int(unet.image_size)
        ctx = getattr(unet, "context_dim", 768)
        out = {}
        for k, n in kinds.items():
            out[k] = [tuple([mk(B, cin, hw, hw)] + [mk(B, 77, ctx) for _ in range(n - 1)]) for _ in range(args.synthetic)]
        return out
    raise SystemExit("give --latents FILE (pre-encoded batches) or --synthetic N: the image / text encoders of the "
                     "reference's loaders are outside this package's scope (SD/train-scripts/_common.py)")
