"""K14 (csrc/salun_tok_bf16.hip): LayerNorm and GEGLU on bf16 tokens against torch's fp32 evaluation of the same
bf16-rounded inputs.  Outputs are rounded to bf16 once: |got - ref| <= 2^-8 |ref| + 1e-5 max|ref|; dgamma / dbeta are
fp32 column sums over up to 32,768 rows: 2e-4 of their scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, want, what, extra=0.0):
    got, want = got.float(), want.float()
    bad = (got - want).abs() > want.abs() * (2.0 ** -8 + extra) + 1e-5 * want.abs().max() + 1e-6
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside, worst {(got - want).abs().max():.3e}"


@pytest.mark.parametrize("shape", [(2, 64, 32), (3, 77, 64), (8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (1, 5, 2048), (2, 3, 8)])
def test_layer_norm_bf16(shape):
    from unlearn_saliency_amd import ops
    B, N, C = shape
    g = torch.Generator(device="cuda").manual_seed(C + N)
    x = (torch.randn(B, N, C, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16)
    ln = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, device="cuda", generator=g) * 0.3 + 1)
        ln.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.2)
    xr = x.float().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), ln.weight, ln.bias, ln.eps)
    xg = x.clone().requires_grad_(True)
    y = ops.layer_norm_bf16(xg, ln)
    assert y.dtype == torch.bfloat16 and y.shape == x.shape
    _close(y, ref, "forward")
    dy = torch.randn(B, N, C, device="cuda", generator=g).to(torch.bfloat16)
    gw, gb, gx = torch.autograd.grad(ref, [ln.weight, ln.bias, xr], dy.float())
    ln.weight.grad = ln.bias.grad = None
    y.backward(dy)
    _close(xg.grad, gx, "dx", extra=2.0 ** -8)  # xhat is rebuilt from bf16 x and fp32 stats: one more rounding-sized term
    assert (ln.weight.grad - gw).abs().max() <= 2e-4 * gw.abs().max() + 1e-5
    assert (ln.bias.grad - gb).abs().max() <= 2e-4 * gb.abs().max() + 1e-5
    y2 = ops.layer_norm_bf16(xg, ln)
    y2.backward(dy)  # accumulates into the existing .grad
    assert (ln.weight.grad - 2 * gw).abs().max() <= 4e-4 * gw.abs().max() + 2e-5


@pytest.mark.parametrize("shape", [(2, 64, 128), (8, 4096, 1280), (8, 64, 5120), (1, 3, 16)])
def test_geglu_bf16(shape):
    from unlearn_saliency_amd import ops
    B, N, Fh = shape
    g = torch.Generator(device="cuda").manual_seed(Fh)
    h = (torch.randn(B, N, 2 * Fh, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    hr = h.float().requires_grad_(True)
    a, b = hr.chunk(2, dim=-1)
    ref = a * F.gelu(b)
    hg = h.clone().requires_grad_(True)
    out = ops.geglu_bf16(hg)
    assert out.shape == (B, N, Fh) and out.dtype == torch.bfloat16
    _close(out, ref, "forward")
    dy = torch.randn(B, N, Fh, device="cuda", generator=g).to(torch.bfloat16)
    (gh,) = torch.autograd.grad(ref, [hr], dy.float())
    out.backward(dy)
    _close(hg.grad, gh, "dh")
