"""The no-grad target pass of the random-label / NSFW-removal steps on a second stream (DDPM
`Diffusion.unlearn_step`, reference runners/diffusion.py:551-566; SD `train_scripts.forget_and_target`, reference
nsfw_removal.py:131-140): issuing it beside the differentiated pass must not change a single bit — same kernels, same
inputs, deterministic reductions — and the SD helper must fall back to the main stream when a weight image had to be
re-packed during the forget pass (cold cache)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from fixtures import ddpm_batch, ddpm_small_config, fill_params, sd_tiny_config
from unlearn_saliency_amd import draws, rng

pytestmark = pytest.mark.gpu


def _dev(shape, seed):
    return torch.from_numpy(rng.normal(int(np.prod(shape)), seed, 0.0, 1.0)).view(*shape).cuda()


def _ddpm_run(overlap, monkeypatch):
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.DDPM.functions import get_optimizer
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    from unlearn_saliency_amd.DDPM.runners import diffusion as RD
    from unlearn_saliency_amd.flat import arena_of
    monkeypatch.setattr(RD, "PSEUDO_OVERLAP", overlap)
    cfg = ddpm_small_config(dropout=0.1)
    model = fill_params(Conditional_Model(cfg), 7000).cuda().train()
    use_salun_convs(model)
    arena = arena_of(model)
    opt = get_optimizer(cfg, arena=arena)
    r = RD.Diffusion.__new__(RD.Diffusion)
    r.args = SimpleNamespace(method="rl", label_to_forget=0, alpha=1e-3)
    r.config, r.device, r.num_timesteps = cfg, torch.device("cuda"), 1000
    r.betas = torch.linspace(1e-4, 0.02, 1000, device="cuda")
    torch.manual_seed(99)
    draws.seed(None)  # dropout keys = f(torch seed, step, call): both runs start at step 0
    losses = []
    for step in range(3):
        rb = tuple(torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in ddpm_batch(4, 300 + step))
        fb = tuple(torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in ddpm_batch(4, 400 + step, label=0))
        rb, fb = (rb[0].float(), rb[1]), (fb[0].float(), fb[1])
        losses.append(r.unlearn_step(model, opt, rb, fb).detach())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), arena.params.clone(), opt.exp_avg.clone()


def test_ddpm_rl_step_identical_with_and_without_overlap(monkeypatch):
    l1, p1, m1 = _ddpm_run(True, monkeypatch)
    l0, p0, m0 = _ddpm_run(False, monkeypatch)
    assert torch.equal(l1, l0), (l1, l0)
    assert torch.equal(p1, p0) and torch.equal(m1, m0)
    assert bool(torch.isfinite(l1).all())


def _sd_model(bf16):
    from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
    m = LatentDiffusionLite(sd_tiny_config(), bf16=bf16)
    fill_params(m.model.diffusion_model, 9000)
    m = m.cuda().train()
    m.use_mfma_convs()  # own kernels in both precisions: deterministic, so the comparisons below can be bit-exact
    return m


@pytest.mark.parametrize("bf16", [False, True])
def test_sd_forget_and_target_identical_and_cold_cache_falls_back(monkeypatch, bf16):
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.SD import train_scripts as TS
    model = _sd_model(bf16)
    cfg = sd_tiny_config()
    B, ctx = 3, cfg["context_dim"]
    z, t = _dev((B, 4, cfg["image_size"], cfg["image_size"]), 5), torch.tensor([3, 500, 998], device="cuda")
    c_f, c_t = _dev((B, 7, ctx), 6), _dev((B, 7, ctx), 7)
    streams_used = []
    real_stream_ctx = torch.cuda.stream

    def spy(s):
        streams_used.append(s)
        return real_stream_ctx(s)

    monkeypatch.setattr(torch.cuda, "stream", spy)
    # cold: nothing packed yet -> (bf16) the forget pass packs, the target pass must stay on the main stream
    packs0 = ops.PACK_CALLS[0]
    out_c, tgt_c = TS.forget_and_target(model, z, t, c_f, c_t)
    cold_streams = len(streams_used)
    if bf16:
        assert ops.PACK_CALLS[0] > packs0 and cold_streams == 0
    # warm: second stream
    out_w, tgt_w = TS.forget_and_target(model, z, t, c_f, c_t)
    assert len(streams_used) == cold_streams + 1
    monkeypatch.setattr(TS, "TARGET_OVERLAP", False)
    out_s, tgt_s = TS.forget_and_target(model, z, t, c_f, c_t)
    torch.cuda.synchronize()
    assert out_w.requires_grad and not tgt_w.requires_grad
    for a in (out_c, out_w):
        assert torch.equal(a, out_s)
    for a in (tgt_c, tgt_w):
        assert torch.equal(a, tgt_s)
    assert not torch.equal(out_s, tgt_s)
    # and the gradient of the step's loss through the overlapped pair is the serial one's
    g = []
    for o, tg in ((out_w, tgt_w), (out_s, tgt_s)):
        model.zero_grad(set_to_none=True)
        ops.mse_loss(tg, o).backward()
        g.append(torch.cat([p.grad.reshape(-1) for p in model.model.diffusion_model.parameters() if p.grad is not None]))
    assert torch.equal(g[0], g[1]) and float(g[0].abs().max()) > 0
