"""SD: the package's U-Net (plain PyTorch, CPU here) against the reference's UNetModel output and parameter tables."""
import os

import numpy as np
import torch

from fixtures import fill_params, sd_tiny_config
from unlearn_saliency_amd import rng


def test_sd_unet_matches_reference(golden_dir):
    from unlearn_saliency_amd.SD.unet import UNetModel, V1_UNET_CONFIG
    g = np.load(os.path.join(golden_dir, "sd_core.npz"))
    m = fill_params(UNetModel(**sd_tiny_config()), 9000).eval()
    assert [n for n, _ in m.named_parameters()] == list(g["tiny_param_names"])
    x = torch.from_numpy(rng.normal(2 * 4 * 8 * 8, 1).reshape(2, 4, 8, 8))
    c = torch.from_numpy(rng.normal(2 * 7 * 24, 2).reshape(2, 7, 24))
    with torch.no_grad():
        out = m(x, torch.tensor([3, 700]), c).numpy()
    ref = g["tiny_forward"]
    assert np.allclose(out, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    with torch.device("meta"):
        full = UNetModel(**V1_UNET_CONFIG)
    assert [n for n, _ in full.named_parameters()] == list(g["full_param_names"])  # 686 mask keys, in order
    assert [str(tuple(p.shape)) for p in full.parameters()] == list(g["full_param_shapes"])
    assert sum(p.numel() for p in full.parameters()) == int(g["full_numel"]) == 859_520_964


def test_ldm_schedule_and_prefix():
    from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
    m = LatentDiffusionLite(sd_tiny_config())
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    ac = np.cumprod(1 - betas)
    assert np.allclose(m.sqrt_alphas_cumprod.numpy(), np.sqrt(ac).astype(np.float32))
    names = [n for n, _ in m.named_parameters()]
    assert all(n.startswith("model.diffusion_model.") for n in names)
    assert names[0].split("model.diffusion_model.")[-1] == "time_embed.0.weight"
