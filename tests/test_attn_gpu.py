"""K13 (csrc/salun_attn.hip): fused attention forward / backward on bf16 tokens against an fp32 evaluation of the
reference's expression (SD/ldm/modules/attention.py:168-192: softmax(scale * q k^T) v) on the same bf16-rounded inputs.

Tolerance: scores and softmax are fp32 on both sides; the kernel rounds P (and dS in the backward) to bf16 before the
second GEMM and the outputs to bf16 — relative 2^-8 each.  Asserted: max |got - ref| <= 2e-2 * max|ref| for o, dq, dk, dv
(measured ~4e-3), and the logsumexp within 1e-4 (absolute, natural-log units)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, H, Nq, Nk, D)
SHAPES = [(2, 4, 64, 64, 8), (2, 4, 16, 7, 16), (1, 2, 100, 77, 40), (2, 8, 256, 256, 40), (1, 8, 1024, 1024, 80),
          (2, 8, 64, 64, 160), (2, 8, 256, 77, 160), (1, 8, 4096, 4096, 40), (3, 8, 130, 77, 80), (1, 1, 1, 1, 40),
          (2, 3, 33, 65, 64), (1, 2, 200, 129, 32)]


@pytest.mark.parametrize("shape", SHAPES)
def test_attention_matches_fp32_reference(shape):
    from unlearn_saliency_amd import ops
    B, H, Nq, Nk, D = shape
    g = torch.Generator(device="cuda").manual_seed(Nq * 7 + Nk)
    mk = lambda n: torch.randn(B, n, H * D, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = mk(Nq).requires_grad_(True), mk(Nk).requires_grad_(True), mk(Nk).requires_grad_(True)
    scale = D ** -0.5
    view = lambda t: t.view(B, t.shape[1], H, D)
    o = ops.attention(view(q), view(k), view(v), scale)
    assert o.shape == (B, Nq, H, D) and o.dtype == torch.bfloat16
    qr, kr, vr = (t.detach().float().view(B, -1, H, D).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bhid,bhjd->bhij", qr, kr) * scale
    ref = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), vr)

    def close(got, want, what):
        err = float((got.float() - want.detach()).abs().max() / want.detach().abs().max().clamp_min(1e-3))
        assert err <= 2e-2, f"{what}: {err:.3e}"
        return err

    e_o = close(o.transpose(1, 2), ref, "o")
    _, lse = ops.attn_forward(view(q.detach()), view(k.detach()), view(v.detach()), scale)
    ref_lse = torch.logsumexp(s.detach(), -1).reshape(B * H, Nq)
    assert (lse * math.log(2.0) - ref_lse).abs().max() <= 1e-4 * max(1.0, float(ref_lse.abs().max()))
    d_o = torch.randn(B, Nq, H, D, device="cuda", generator=g).to(torch.bfloat16)
    o.backward(d_o)
    ref.backward(d_o.float().transpose(1, 2))
    e_q = close(q.grad.view(B, Nq, H, D).transpose(1, 2), qr.grad, "dq")
    e_k = close(k.grad.view(B, Nk, H, D).transpose(1, 2), kr.grad, "dk")
    e_v = close(v.grad.view(B, Nk, H, D).transpose(1, 2), vr.grad, "dv")
    print(f"{shape}: o {e_o:.1e} dq {e_q:.1e} dk {e_k:.1e} dv {e_v:.1e}")


def test_strided_views_of_a_fused_projection():
    """q / k / v as column slices of one [B, N, 3*H*D] projection (token stride 3*H*D): read in place."""
    from unlearn_saliency_amd import ops
    B, H, N, D = 2, 8, 96, 40
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn(B, N, 3 * H * D, device="cuda", generator=g).to(torch.bfloat16)
    parts = [qkv[..., i * H * D:(i + 1) * H * D].view(B, N, H, D) for i in range(3)]
    o, _ = ops.attn_forward(*parts, D ** -0.5, need_lse=False)
    oc, _ = ops.attn_forward(*[p.contiguous() for p in parts], D ** -0.5, need_lse=False)
    assert torch.equal(o, oc)


@pytest.mark.parametrize("scale", [0.0, -0.125, float("inf"), float("nan")])
def test_non_positive_or_non_finite_scale_is_refused(scale):
    """the kernels take the row maximum on the raw scores and apply scale*log2(e) afterwards, which is only
    monotone for a positive finite scale; anything else must be an error, not inf / NaN outputs."""
    from unlearn_saliency_amd import ops
    q = torch.randn(1, 16, 2, 40, device="cuda").to(torch.bfloat16)
    with pytest.raises(Exception) as ei:
        ops.attention(q, q, q, scale)
    assert "EINVAL" in str(ei.value) or "-22" in str(ei.value) or "invalid" in str(ei.value).lower()
