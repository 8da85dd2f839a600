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


def test_compvis_to_diffusers_key_layout_matches_reference(golden_dir):
    """SD/convert.py against the REFERENCE's convert_ldm_unet_checkpoint (convertModels.py:348-591): same keys and the
    same tensor behind every key on the tiny configuration; same key set and shapes on the full v1 U-Net."""
    from unlearn_saliency_amd.SD.convert import convert_ldm_unet_checkpoint
    from unlearn_saliency_amd.SD.unet import UNetModel, V1_UNET_CONFIG
    g = np.load(os.path.join(golden_dir, "sd_convert.npz"))
    cfg = sd_tiny_config()
    m = fill_params(UNetModel(**cfg), 9000)
    sd = {"model.diffusion_model." + k: v for k, v in m.state_dict().items()}
    sd["first_stage_model.ignored"] = torch.zeros(1)
    conv = convert_ldm_unet_checkpoint(sd, cfg["num_res_blocks"])
    ref = dict(zip(g["tiny_keys"], zip(g["tiny_sums"], g["tiny_shapes"])))
    assert set(conv) == set(ref) and len(conv) == len(ref)
    for k, v in conv.items():
        assert str(tuple(v.shape)) == ref[k][1], k
        assert abs(float(v.double().sum()) - float(ref[k][0])) <= 1e-9 * max(1.0, abs(float(ref[k][0]))), k
    with torch.device("meta"):
        full = UNetModel(**V1_UNET_CONFIG)
    convf = convert_ldm_unet_checkpoint(full.state_dict(), V1_UNET_CONFIG["num_res_blocks"])
    assert sorted(convf) == list(g["full_keys_sorted"])
    assert [str(tuple(convf[k].shape)) for k in sorted(convf)] == list(g["full_shapes_sorted"])
