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


def make_convert():
    """Key layout produced by the REFERENCE's convert_ldm_unet_checkpoint (SD/train-scripts/convertModels.py:348-591;
    diffusers / omegaconf are stubbed: the function only renames keys): tiny config with per-tensor checksums, full
    v1 config as the ordered key list."""
    import types
    from types import SimpleNamespace
    for m in ("omegaconf", "diffusers", "diffusers.pipelines", "diffusers.pipelines.latent_diffusion",
              "diffusers.pipelines.latent_diffusion.pipeline_latent_diffusion", "diffusers.pipelines.paint_by_example",
              "diffusers.pipelines.stable_diffusion"):
        _stub(m)
    sys.modules["omegaconf"].OmegaConf = type("OmegaConf", (), {})

    class _Any(types.ModuleType):
        def __getattr__(self, n):
            return type(n, (), {})
    for m in list(sys.modules):
        if m == "diffusers" or m.startswith("diffusers."):
            sys.modules[m].__class__ = _Any
    sys.path.insert(0, "/root/reference/SD/train-scripts")
    import convertModels as CM
    from unlearn_saliency_amd.SD.unet import UNetModel, V1_UNET_CONFIG
    out = {}
    cfg = sd_tiny_config()
    m = fill_params(UNetModel(**cfg), 9000)
    sd = {"model.diffusion_model." + k: v for k, v in m.state_dict().items()}
    conv = CM.convert_ldm_unet_checkpoint(dict(sd), {"layers_per_block": cfg["num_res_blocks"]})
    out["tiny_keys"] = np.array(list(conv.keys()))
    out["tiny_sums"] = np.array([float(v.double().sum()) for v in conv.values()])
    out["tiny_shapes"] = np.array([str(tuple(v.shape)) for v in conv.values()])
    with torch.device("meta"):
        full = UNetModel(**V1_UNET_CONFIG)
    sdf = {"model.diffusion_model." + k: v for k, v in full.state_dict().items()}
    convf = CM.convert_ldm_unet_checkpoint(dict(sdf), {"layers_per_block": V1_UNET_CONFIG["num_res_blocks"]})
    out["full_keys_sorted"] = np.array(sorted(convf.keys()))
    out["full_shapes_sorted"] = np.array([str(tuple(convf[k].shape)) for k in sorted(convf.keys())])
    np.savez_compressed(os.path.join(HERE, "sd_convert.npz"), **out)
    print("sd convert fixtures written:", len(conv), "tiny keys,", len(convf), "full keys")


if __name__ == "__main__":
    make_sd()
    make_convert()
