"""SD golden vectors: the reference's `UNetModel` (imported from /root/reference/SD, build container only;
omegaconf is stubbed — it is only touched for an isinstance check) on generator-filled weights."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
from fixtures import fill_params, sd_tiny_config  # noqa: E402
from make_golden import _stub  # noqa: E402
from unlearn_saliency_amd import rng  # noqa: E402


def make_sd():
    _stub("omegaconf")
    _stub("omegaconf.listconfig")
    sys.modules["omegaconf.listconfig"].ListConfig = type("ListConfig", (), {})
    sys.path.insert(0, "/root/reference/SD")
    from ldm.modules.diffusionmodules.openaimodel import UNetModel as RefUNet
    from unlearn_saliency_amd.SD.unet import V1_UNET_CONFIG
    out = {}
    cfg = sd_tiny_config()
    m = fill_params(RefUNet(**cfg), 9000).eval()
    x = torch.from_numpy(rng.normal(2 * 4 * 8 * 8, 1).reshape(2, 4, 8, 8))
    t = torch.tensor([3, 700])
    c = torch.from_numpy(rng.normal(2 * 7 * 24, 2).reshape(2, 7, 24))
    with torch.no_grad():
        out["tiny_forward"] = m(x, t, c).numpy()
    out["tiny_param_names"] = np.array([n for n, _ in m.named_parameters()])
    with torch.device("meta"):
        full = RefUNet(**V1_UNET_CONFIG)
    out["full_param_names"] = np.array([n for n, _ in full.named_parameters()])
    out["full_param_shapes"] = np.array([str(tuple(p.shape)) for p in full.parameters()])
    out["full_numel"] = np.int64(sum(p.numel() for p in full.parameters()))
    np.savez_compressed(os.path.join(HERE, "sd_core.npz"), **out)
    print("sd fixtures written")


if __name__ == "__main__":
    make_sd()
