"""BASELINE.json configs[3] and [4] at FULL model size on the device: one unlearning step
of the CFG-DDPM U-Net (38,632,323 parameters, batch 128) and one of the SD-v1 U-Net in its bf16 configuration
(859,520,964 parameters, batch 8, 64x64 latents) — the configurations `bench.py --workload ddpm | sd` time — with the
size-independent properties the path offers: finite loss, masked-out weights bit-identical to their initial values,
selected weights moved, Adam moments zero where the mask is zero, and NO library convolution anywhere in the step.
(The arithmetic itself is pinned on the reduced U-Nets against reference-run goldens: tests/test_ddpm_gpu.py,
tests/test_sd_parity_gpu.py.)"""
import os
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _saliency_mask(n, seed):
    from unlearn_saliency_amd import ops
    sal = ops.fill_normal(n, seed, 0.0, 1e-3) * (1.0 + ops.fill_uniform(n, seed + 1, 0.0, 0.5))
    m = ops.mask_topk(sal, [int(n * 0.5)], check=True)[0]
    assert ops.mask_popcount(m) == int(n * 0.5)
    return m


def _check_masked_update(arena, opt, mask, theta0):
    frozen = mask == 0
    assert torch.equal(arena.params[frozen], theta0[frozen]), "a masked-out weight moved"
    moved = (arena.params[~frozen] != theta0[~frozen]).float().mean()
    assert float(moved) > 0.9, float(moved)
    assert not opt.exp_avg[frozen].any() and not opt.exp_avg_sq[frozen].any()
    assert bool(torch.isfinite(arena.params).all())


def test_ddpm_full_size_unlearn_step(capsys):
    from unlearn_saliency_amd import conv as sconv
    from unlearn_saliency_amd.DDPM.functions import get_optimizer, load_config
    from unlearn_saliency_amd.DDPM.runners.diffusion import Diffusion
    from unlearn_saliency_amd.flat import arena_of
    cfg = load_config(os.path.join(ROOT, "unlearn_saliency_amd", "DDPM", "configs", "cifar10_saliency_unlearn.yml"))
    assert cfg.training.batch_size == 128
    args = SimpleNamespace(ckpt_folder=None, label_to_forget=0, cond_scale=2.0, mask_path=None, method="rl",
                           alpha=1e-3, synthetic=True, library_conv=False)
    torch.manual_seed(1234)
    runner = Diffusion(args, cfg)
    remain_loader, forget_loader = runner._loaders()
    model = runner._load_model()
    capsys.readouterr()
    arena = arena_of(model)
    assert arena.n == 38_632_323
    mask = _saliency_mask(arena.n, 5)
    opt = get_optimizer(cfg, arena=arena)
    opt.set_mask(mask)
    theta0 = arena.params.clone()
    model.train()
    sconv.reset_library_conv_calls()
    rb, fb = next(iter(remain_loader)), next(iter(forget_loader))
    assert rb[0].shape == (128, 3, 32, 32) and bool((fb[1] == 0).all()) and not bool((rb[1] == 0).any())
    loss = runner.unlearn_step(model, opt, rb, fb)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and float(loss) > 0
    assert sconv.library_conv_calls() == 0, sconv.LIBRARY_CONV_CALLS
    _check_masked_update(arena, opt, mask, theta0)
    # Phase A on the same model: one forget batch through the CFG loss, clipped and accumulated
    acc = runner.accumulate_saliency(model, [fb], arena)
    assert bool(torch.isfinite(acc).all()) and float(acc.abs().max()) > 0
    assert float(acc.double().norm()) <= 1.0 + 1e-4  # one batch, clipped to norm 1 (runners/diffusion.py:985-990)
    assert sconv.library_conv_calls() == 0, sconv.LIBRARY_CONV_CALLS


def test_sd_v1_bf16_full_size_unlearn_step():
    from unlearn_saliency_amd import conv as sconv
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.optim import FusedMaskedAdam
    from unlearn_saliency_amd.SD import train_scripts as TS
    from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = LatentDiffusionLite(bf16=True).to(dev)
    arena = TS._unet_arena(model)
    assert arena.n == 859_520_964
    assert model.use_mfma_convs() == 96       # every convolution but the 4-channel head / tail on the bf16 kernels
    assert model.fill_zero_initialised() > 1_000_000  # zero_module layers + biases: a live network, like a checkpoint
    mask = _saliency_mask(arena.n, 7)
    opt = FusedMaskedAdam(arena, lr=1e-5)
    opt.set_mask(mask)
    theta0 = arena.params.clone()
    model.train()
    B = 8
    mk = lambda *s: torch.randn(*s, device=dev)
    z_f, c_f, c_p, z_r, c_r = mk(B, 4, 64, 64), mk(B, 77, 768), mk(B, 77, 768), mk(B, 4, 64, 64), mk(B, 77, 768)
    sconv.reset_library_conv_calls()
    # the loop body of nsfw_removal (SD/train-scripts/nsfw_removal.py:88-160; SD/train_scripts.py::_unlearn)
    opt.zero_grad()
    remain_loss = model.shared_step({"z": z_r, "c": c_r})[0]
    t = torch.randint(0, model.num_timesteps, (B,), device=dev).long()
    noise = torch.randn_like(z_f)
    z_noisy = model.q_sample(x_start=z_f, t=t, noise=noise)
    forget_out, pseudo_out = TS.forget_and_target(model, z_noisy, t, c_f, c_p)  # target pass on a second stream
    loss = ops.mse_loss(pseudo_out, forget_out) + 0.1 * remain_loss
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and float(loss) > 0
    assert sconv.library_conv_calls() == 0, sconv.LIBRARY_CONV_CALLS
    assert all(p.dtype == torch.float32 for p in model.model.diffusion_model.parameters())  # fp32 master weights
    _check_masked_update(arena, opt, mask, theta0)


def test_side_stream_backward_weight_never_changes_a_gradient_bit():
    """Full-size CFG-DDPM U-Net, batch 128: every parameter gradient with backward-weight on the side stream equals,
    bit for bit and run after run, the single-stream result.  Round 4 found the one way it could differ: autograd
    accumulates the residual's second contribution INTO the gradient buffer (in place, main stream) that the first
    AttnBlock's proj_out backward-weight kernel was still reading on the side stream — visible once the own fp32
    attention made that block's backward short (resblock.hold_until_join)."""
    from unlearn_saliency_amd import resblock
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.DDPM.functions import load_config
    from unlearn_saliency_amd.DDPM.functions.losses import loss_registry_conditional
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    from unlearn_saliency_amd.flat import arena_of
    cfg = load_config(os.path.join(ROOT, "unlearn_saliency_amd", "DDPM", "configs", "cifar10_saliency_unlearn.yml"))
    torch.manual_seed(0)
    model = Conditional_Model(cfg).cuda().train()
    assert use_salun_convs(model) > 0 and model.own_gemm
    arena = arena_of(model)
    x = torch.rand(128, 3, 32, 32, device="cuda") * 2 - 1
    c = torch.randint(0, 10, (128,), device="cuda")
    e, t = torch.randn_like(x), torch.randint(0, 1000, (128,), device="cuda")
    b = torch.linspace(1e-4, 0.02, 1000, device="cuda")

    from unlearn_saliency_amd import ops

    def grads(overlap):
        prev = resblock.OVERLAP_WGRAD
        resblock.OVERLAP_WGRAD = overlap
        ops.WGRAD_KERNEL[0] = "shared"   # the same backward-weight kernel on either schedule (round 6: a launch that does
        try:                             # not share the device takes the ring kernel — another summation order)
            torch.manual_seed(5)  # label drop
            from unlearn_saliency_amd import draws
            draws.set_state((5, 1, 0))  # dropout keys
            arena.zero_grad()
            loss_registry_conditional["simple"](model, x, t, c, e, b).backward()
            torch.cuda.synchronize()
        finally:
            resblock.OVERLAP_WGRAD = prev
            ops.WGRAD_KERNEL[0] = None
        return arena.grads.clone()

    ref = grads(False)
    assert torch.equal(grads(False), ref)
    for rep in range(6):
        g = grads(True)
        bad = [n for n, o, k in zip(arena.names, arena.offsets, arena.numels) if not torch.equal(g[o:o + k], ref[o:o + k])]
        assert not bad, (rep, bad[:5])
