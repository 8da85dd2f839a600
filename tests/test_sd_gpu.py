"""SD scripts on the GPU (tiny U-Net config): `generate_mask` / `certain_label` vs a plain-PyTorch restatement of
the reference loops (per-tensor grads, torch.optim.Adam, per-tensor `p.grad *= mask[...]`,
SD/train-scripts/generate_mask.py:24-108, random_label.py:58-139) on identical inputs and random draws."""
import os

import numpy as np
import pytest
import torch

from fixtures import fill_params, sd_tiny_config
from unlearn_saliency_amd import rng

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _batches(nb, seed, with_pseudo=False):
    out = []
    for b in range(nb):
        z = _t(rng.normal(4 * 4 * 8 * 8, seed + 10 * b).reshape(4, 4, 8, 8))
        c1 = _t(rng.normal(4 * 7 * 24, seed + 10 * b + 1).reshape(4, 7, 24))
        c2 = _t(rng.normal(4 * 7 * 24, seed + 10 * b + 2).reshape(4, 7, 24))
        out.append((z, c1, c2) if with_pseudo else (z, c1))
    return out


class _Replay:
    """Make torch.randint / torch.randn_like deterministic and identical for both implementations."""

    def __init__(self, seed):
        self.g = torch.Generator(device="cuda").manual_seed(seed)

    def __enter__(self):
        self.real = (torch.randint, torch.randn_like)
        g = self.g
        torch.randint = lambda lo, hi, size, device=None, **k: self.real[0](lo, hi, size, device="cuda", generator=g)
        torch.randn_like = lambda x, **k: torch.randn(x.shape, device=x.device, generator=g)
        return self

    def __exit__(self, *a):
        torch.randint, torch.randn_like = self.real


def _model():
    from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
    m = LatentDiffusionLite(sd_tiny_config())
    fill_params(m.model.diffusion_model, 9000)
    return m.cuda()


def test_generate_mask_matches_plain_torch(tmp_path, monkeypatch):
    from unlearn_saliency_amd.SD import train_scripts as TS
    monkeypatch.chdir(tmp_path)
    batches = _batches(2, 100, with_pseudo=True)
    m1 = _model()
    with _Replay(5):
        mask = TS.generate_mask("3", 7.5, 4, 1, 1e-5, None, None, None, "cuda", model=m1, forget_dl=batches)
    saved = torch.load(tmp_path / "mask" / "3" / "with_0.5.pt", weights_only=False)
    unet_names = [n for n, _ in m1.model.diffusion_model.named_parameters()]
    assert list(saved.keys()) == unet_names and all(v.dtype == torch.int64 for v in saved.values())
    # plain restatement
    m2 = _model()
    m2.eval()
    grads = {n: 0 for n in unet_names}
    with _Replay(5):
        for z, c, c0 in batches:
            m2.zero_grad()
            t = torch.randint(0, m2.num_timesteps, (z.shape[0],), device="cuda").long()
            noise = torch.randn_like(z)
            zn = m2.q_sample(z, t, noise)
            preds = (1 + 7.5) * m2.apply_model(zn, t, c) - 7.5 * m2.apply_model(zn, t, c0)
            (-torch.nn.MSELoss()(noise, preds)).backward()
            for n, p in m2.model.diffusion_model.named_parameters():
                if p.grad is not None:
                    grads[n] = grads[n] + p.grad.detach().clone()
    flat = torch.cat([grads[n].abs().flatten() for n in unet_names])
    k = int(flat.numel() * 0.5)
    ranks = torch.argsort(torch.argsort(-flat, stable=True), stable=True)
    ref = (ranks < k).to(torch.uint8)
    assert int(mask.sum()) == k
    assert (mask != ref).float().mean() < 2e-3  # accumulators agree to fp32 rounding; only near-threshold flips


@pytest.mark.parametrize("method", ["full", "xattn"])
def test_certain_label_matches_plain_torch(tmp_path, method):
    from unlearn_saliency_amd.SD import train_scripts as TS
    forget = _batches(2, 200, with_pseudo=True)
    remain = _batches(2, 300)
    m1 = _model()
    names = [n for n, _ in m1.model.diffusion_model.named_parameters()]
    sizes = [p.numel() for p in m1.model.diffusion_model.parameters()]
    n = sum(sizes)
    mflat = (rng.u8(n, 77) & 1).astype(np.int64)
    off = np.cumsum([0] + sizes)
    mask = {k: torch.from_numpy(mflat[off[i]:off[i + 1]]).view_as(p)
            for i, (k, p) in enumerate(m1.model.diffusion_model.named_parameters())}
    mpath = tmp_path / "mask.pt"
    torch.save(mask, mpath)
    with _Replay(9):
        _, losses = TS.certain_label(3, method, 0.5, 4, 1, 1e-4, None, None, str(mpath), None, "cuda", model=m1,
                                     forget_dl=forget, remain_dl=remain)
    # plain restatement (reference loop order: remain loss, forget/pseudo, backward, mask multiply, Adam)
    m2 = _model()
    m2.train()
    params = [p for k, p in m2.model.diffusion_model.named_parameters() if method == "full" or "attn2" in k]
    opt = torch.optim.Adam(params, lr=1e-4)
    ref_losses = []
    with _Replay(9):
        for (zf, cf, cp), (zr, cr) in zip(forget, remain):
            opt.zero_grad()
            m2.zero_grad()
            t = torch.randint(0, m2.num_timesteps, (zr.shape[0],), device="cuda").long()
            noise = torch.randn_like(zr)
            remain_loss = torch.nn.MSELoss()(noise, m2.apply_model(m2.q_sample(zr, t, noise), t, cr))
            t = torch.randint(0, m2.num_timesteps, (zf.shape[0],), device="cuda").long()
            noise = torch.randn_like(zf)
            zn = m2.q_sample(zf, t, noise)
            out = m2.apply_model(zn, t, cf)
            pseudo = m2.apply_model(zn, t, cp).detach()
            loss = torch.nn.MSELoss()(out, pseudo) + 0.5 * remain_loss
            loss.backward()
            for k, p in m2.named_parameters():
                if p.grad is not None:
                    p.grad *= mask[k.split("model.diffusion_model.")[-1]].to("cuda")
            opt.step()
            ref_losses.append(loss.item())
    assert np.allclose(losses, ref_losses, rtol=1e-4)
    a = torch.cat([p.detach().flatten() for p in m1.model.diffusion_model.parameters()]).cpu().numpy()
    b = torch.cat([p.detach().flatten() for p in m2.model.diffusion_model.parameters()]).cpu().numpy()
    init = torch.cat([p.detach().flatten() for p in _model().model.diffusion_model.parameters()]).cpu().numpy()
    assert np.array_equal(a[mflat == 0], init[mflat == 0])  # masked-out weights untouched
    lr = 1e-4
    close = np.abs(a - b) <= 0.02 * lr + 1e-6 * np.abs(b)
    assert close.mean() > 0.99
    moved = a != init
    if method == "xattn":
        sel = np.concatenate([np.full(s, "attn2" in k) for k, s in zip(names, sizes)])
        assert not moved[~sel].any() and moved[sel & (mflat == 1)].mean() > 0.9


@pytest.mark.parametrize("bf16", [False, True])
def test_resident_activations_give_the_checkpointed_gradient_bit_for_bit(bf16):
    """unet.set_activation_checkpointing(False) (bench `resident_activations`, `--resident_activations` of the command
    lines): the reference's config re-runs every block inside backward to save memory; keeping the activations instead
    changes no kernel and no order of accumulation — the flat gradient of a remain + forget pass is identical."""
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.SD import train_scripts as TS
    from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
    from unlearn_saliency_amd.SD.unet import set_activation_checkpointing
    cfg = dict(sd_tiny_config(), use_checkpoint=True)
    m = LatentDiffusionLite(cfg, bf16=bf16)
    fill_params(m.model.diffusion_model, 9100)
    m = m.cuda()
    m.use_mfma_convs()
    arena = TS._unet_arena(m)
    m.train()
    (z_r, c_r), (z_f, c_f, c_p) = _batches(1, 500)[0], _batches(1, 600, with_pseudo=True)[0]
    t = torch.tensor([5, 400, 77, 901], device="cuda")
    noise = _t(rng.normal(4 * 4 * 8 * 8, 700).reshape(4, 4, 8, 8))

    def grad():
        arena.zero_grad()
        with _Replay(11):
            remain = m.shared_step({"z": z_r, "c": c_r})[0]
        z_noisy = m.q_sample(x_start=z_f, t=t, noise=noise)
        fo, po = TS.forget_and_target(m, z_noisy, t, c_f, c_p)
        (ops.mse_loss(po, fo) + 0.1 * remain).backward()
        torch.cuda.synchronize()
        return arena.grads.clone()

    g_ckpt = grad()
    assert float(g_ckpt.abs().max()) > 0
    assert set_activation_checkpointing(m.model.diffusion_model, False) > 0
    g_res = grad()
    assert torch.equal(g_ckpt, g_res)
    assert set_activation_checkpointing(m.model.diffusion_model, True) > 0
    assert torch.equal(grad(), g_ckpt)
