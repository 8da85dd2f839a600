"""K15 (csrc/salun_gemm.hip, gemm.py): the fp32 MFMA GEMM behind the diffusion U-Nets' Linear layers and fp32 attention,
through the C-ABI, against float64 evaluations of the library ops it replaces (nn.Linear, bmm -> softmax -> bmm:
reference DDPM/models/diffusion.py:85-192, SD/ldm/modules/attention.py:149-200).

`v_mfma_f32_32x32x2_f32` is an exact fp32 FMA chain, so the only difference to any other fp32 GEMM is summation order:
tolerances are a few 1e-6 of each tensor's scale (the measured worst is printed and logged)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from unlearn_saliency_amd import rng

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(shape, seed, std=1.0):
    return torch.from_numpy(rng.normal(int(np.prod(shape)), seed, 0.0, std)).view(*shape).cuda()


def rel(got, want):
    want = want.double()
    return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def log(name, err):
    print(f"{name}: {err:.3e}")
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "r04_measured_errors.txt"), "a") as f:
            f.write(f"{name} {err:.4e}\n")


@pytest.mark.parametrize("M,N,K", [(128, 512, 128), (128, 256, 1024), (130, 70, 50), (256, 10, 512), (1, 5, 3),
                                   (4096, 320, 320), (64, 64, 4096), (300, 1280, 77)])
def test_mm_nt_matches_float64(M, N, K):
    from unlearn_saliency_amd import gemm
    a, b, bias = dev((M, K), 1), dev((N, K), 2), dev((N,), 3)
    want = a.double() @ b.double().t() + bias.double()
    got = gemm.mm_nt(a, b, bias=bias)
    e = rel(got, want)
    log(f"mm_nt {M}x{N}x{K}", e)
    assert e <= 3e-6, e
    # deterministic (the split-K fold has a fixed order, no atomics)
    assert torch.equal(gemm.mm_nt(a, b, bias=bias), got)
    # accumulate + alpha into a strided output
    out = torch.zeros(M, N + 3, device="cuda")
    base = dev((M, N), 4)
    out[:, :N] = base
    gemm.mm_nt(a, b, out=out[:, :N], alpha=0.5, accumulate=True)
    e = rel(out[:, :N], base.double() + 0.5 * (a.double() @ b.double().t()))
    assert e <= 3e-6 and bool((out[:, N:] == 0).all()), e


def test_every_operand_orientation_and_the_scalar_path():
    """x.W^T, dY.W, dY^T.x of a Linear layer are ONE kernel with different strides; so are misaligned views."""
    from unlearn_saliency_amd import gemm
    M, N, K = 200, 96, 136
    for a_t in (False, True):
        for b_t in (False, True):
            a = dev((K, M), 5).t() if a_t else dev((M, K), 5)       # a_t: row index contiguous
            b = dev((K, N), 6).t() if b_t else dev((N, K), 6)
            e = rel(gemm.mm_nt(a, b), a.double() @ b.double().t())
            log(f"mm_nt orientation a_t={a_t} b_t={b_t}", e)
            assert e <= 3e-6, (a_t, b_t, e)
    # neither stride is 1 (every second column of a wider matrix) and a base pointer off the 16-byte grid
    a = dev((M, 2 * K), 7)[:, ::2]
    b = dev((N * K + 1,), 8)[1:].view(N, K)
    e = rel(gemm.mm_nt(a, b), a.double() @ b.double().t())
    assert e <= 3e-6, e


def test_linear_forward_backward_and_gradient_sinks():
    from unlearn_saliency_amd import gemm
    from unlearn_saliency_amd.flat import arena_of
    torch.manual_seed(0)
    lin = torch.nn.Linear(1024, 256).cuda()
    ref = torch.nn.Linear(1024, 256).double().cuda()
    ref.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    assert gemm.use_salun_linears(lin) == 1 and type(lin) is gemm.SalunLinear
    arena = arena_of(lin)  # .grad = views of the flat gradient: the kernels accumulate into them
    x = dev((8, 16, 1024), 11).requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    dy = dev((8, 16, 256), 12)
    for step in range(2):  # twice: the second pass must ADD into the sinks
        lin(x).backward(dy)
        ref(xr).backward(dy.double())
    y = lin(x)
    worst = max(rel(y, ref(xr)), rel(x.grad, xr.grad), rel(lin.weight.grad, ref.weight.grad), rel(lin.bias.grad, ref.bias.grad))
    log("linear 128x1024->256 fwd / dx / dW / db (2 accumulated passes)", worst)
    assert worst <= 3e-6, worst
    assert lin.weight.grad.data_ptr() == arena.grads.data_ptr()  # still the arena's view: nothing re-pointed .grad


def test_grouped_linear_equals_separate_layers():
    """The DDPM's per-block embedding projections: 22 Linear(1024 -> 128 | 256) on one [128, 1024] activation."""
    from unlearn_saliency_amd import gemm
    torch.manual_seed(1)
    widths = [128] * 6 + [256] * 16
    layers = [torch.nn.Linear(1024, w).cuda() for w in widths]
    x = dev((128, 1024), 21).requires_grad_(True)
    outs = gemm.grouped_linear(x, layers)
    assert len(outs) == 22
    gs = [dev((128, w), 30 + i) for i, w in enumerate(widths)]
    gs[3] = None  # a projection whose block saw no gradient
    torch.autograd.backward([o for o, g in zip(outs, gs) if g is not None], [g for g in gs if g is not None])
    xr = x.detach().double().requires_grad_(True)
    worst = 0.0
    refs = []
    for l, o in zip(layers, outs):
        r = F.linear(xr, l.weight.detach().double().requires_grad_(True), l.bias.detach().double().requires_grad_(True))
        refs.append(r)
        worst = max(worst, rel(o, r))
    wr = [(l.weight.detach().double().requires_grad_(True), l.bias.detach().double().requires_grad_(True)) for l in layers]
    tot = sum((F.linear(xr, w, b) * g.double()).sum() for (w, b), g in zip(wr, gs) if g is not None)
    tot.backward()
    worst = max(worst, rel(x.grad, xr.grad))
    for i, (l, (w, b)) in enumerate(zip(layers, wr)):
        if gs[i] is None:
            assert l.weight.grad is None or float(l.weight.grad.abs().max()) == 0
            continue
        worst = max(worst, rel(l.weight.grad, w.grad), rel(l.bias.grad, b.grad))
    log("grouped_linear 22 x (1024 -> 128|256), batch 128", worst)
    assert worst <= 3e-6, worst


def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)


@pytest.mark.parametrize("layout,B,H,Tq,Tk,D", [("ddpm", 8, 1, 256, 256, 256), ("ddpm", 3, 1, 64, 64, 128),
                                                ("sd", 2, 8, 256, 256, 40), ("sd", 2, 8, 64, 77, 160),
                                                ("sd", 1, 4, 1024, 1024, 80)])
def test_attention_f32_matches_float64_in_both_layouts(layout, B, H, Tq, Tk, D):
    """DDPM AttnBlock: channel-major [B, C, T] memory viewed as [B, 1, T, C]; SD: [b, n, h*d] viewed as [b, h, n, d]."""
    from unlearn_saliency_amd import gemm

    def mk(T, seed):
        if layout == "ddpm":
            return dev((B, D, T), seed).view(B, 1, D, T).transpose(2, 3)
        return dev((B, T, H * D), seed).view(B, T, H, D).transpose(1, 2)

    q, k, v = (mk(Tq, 41).requires_grad_(True), mk(Tk, 42).requires_grad_(True), mk(Tk, 43).requires_grad_(True))
    scale = D ** -0.5
    assert gemm.attention_supported(q, k, v) == (Tq * Tk <= gemm.ATTENTION_MAX_SCORES)  # the model-side routing rule
    o = gemm.attention_f32(q, k, v, scale)
    assert o.stride() == q.stride()  # written in the caller's layout: no transposing copy on the way back
    do = mk(Tq, 44)
    o.backward(do)
    qr, kr, vr = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    orf = _attn_ref(qr, kr, vr, scale)
    orf.backward(do.double())
    errs = dict(o=rel(o, orf), dq=rel(q.grad, qr.grad), dk=rel(k.grad, kr.grad), dv=rel(v.grad, vr.grad))
    log(f"attention_f32 {layout} B{B} H{H} {Tq}x{Tk} d{D}", max(errs.values()))
    assert max(errs.values()) <= 5e-6, errs


def test_softmax_rows_and_backward():
    from unlearn_saliency_amd import gemm
    s = dev((3, 5, 77), 51, 4.0)
    want = s.double().softmax(-1)
    p = gemm.softmax_rows_(s.clone())
    assert rel(p, want) <= 1e-6
    dp = dev((3, 5, 77), 52)
    want_ds = 0.125 * want * (dp.double() - (dp.double() * want).sum(-1, keepdim=True))
    got = gemm.softmax_rows_backward_(p, dp.clone(), 0.125)
    assert rel(got, want_ds) <= 2e-6


def test_ddpm_unet_gradients_with_own_gemm_on_vs_off():
    """Reduced CFG-DDPM U-Net: loss and every parameter gradient with the Linear layers / attention on K15 (projections
    grouped) against the same network with them on the library."""
    from fixtures import ddpm_batch, ddpm_small_config, fill_params
    from unlearn_saliency_amd import conv as sconv
    from unlearn_saliency_amd.DDPM.functions.losses import loss_registry_conditional
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    from unlearn_saliency_amd.flat import arena_of
    cfg = ddpm_small_config()
    outs = []
    for own in (True, False):
        sconv._OWN_GEMM[0] = own
        try:
            model = fill_params(Conditional_Model(cfg), 7000).cuda().train()
            assert sconv.use_salun_convs(model) > 0
        finally:
            sconv._OWN_GEMM[0] = True
        assert model.own_gemm is own
        arena = arena_of(model)
        x, c = ddpm_batch(6, 300)
        x, c = torch.from_numpy(x).float().cuda(), torch.from_numpy(c).cuda()
        e = dev(tuple(x.shape), 61)
        t = torch.tensor([3, 500, 998, 17, 250, 750], device="cuda")
        b = torch.linspace(1e-4, 0.02, 1000, device="cuda")
        arena.zero_grad()
        loss = loss_registry_conditional["simple"](model, 2 * x - 1, t, c, e, b, cond_drop_prob=0.0)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((float(loss), arena.grads.clone(), arena))
    (l1, g1, arena), (l0, g0, _) = outs
    assert abs(l1 - l0) <= 2e-6 * abs(l0), (l1, l0)
    worst = 0.0
    gmax = float(g0.abs().max())
    for name, off, k in zip(arena.names, arena.offsets, arena.numels):
        a, b_ = g1[off:off + k], g0[off:off + k]
        if name.endswith(".k.bias"):
            # a key bias shifts every score of a query row by the same amount: softmax does not see it, the true gradient is
            # exactly 0 and both routes deliver their own round-off — checked as "negligible", not against each other
            assert float(a.abs().max()) <= 1e-6 * gmax and float(b_.abs().max()) <= 1e-6 * gmax, name
            continue
        scale = max(float(b_.abs().max()), 1e-9 * gmax)  # (a parameter no output depends on has a zero gradient on both routes)
        err = float((a - b_).abs().max()) / scale
        if err > 2e-5:
            print(f"  {name}: {err:.3e} (scale {scale:.3e})")
        worst = max(worst, err)
    log("DDPM reduced U-Net, own GEMM on vs off: worst parameter gradient", worst)
    assert worst <= 2e-5, worst
