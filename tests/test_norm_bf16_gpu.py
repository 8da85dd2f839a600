"""K12 (csrc/salun_norm_bf16.hip): GroupNorm (+ SiLU) on bf16 NHWC activations against torch's fp32 group_norm on the
same bf16-rounded input.  Tolerance: the kernel computes in fp32 from the bf16 input and rounds the result to bf16 once,
so forward / dx agree with the fp32 reference to one bf16 rounding (2^-8 relative) plus 1e-5 of the tensor's scale;
dgamma / dbeta are fp32 sums: 1e-4 relative to their scale (fp32 partials over <= 64 chunks, folded in fp64)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, C, H, W, groups)
SHAPES = [(2, 32, 8, 8, 32), (2, 64, 4, 4, 32), (3, 96, 5, 7, 32), (2, 320, 32, 32, 32), (2, 960, 16, 16, 32),
          (8, 1280, 8, 8, 32), (1, 2560, 8, 8, 32), (2, 64, 1, 1, 32)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("silu", [True, False])
def test_gn_bf16_matches_fp32_reference(shape, silu):
    from unlearn_saliency_amd import norm
    N, C, H, W, G = shape
    g = torch.Generator(device="cuda").manual_seed(C + H)
    x = (torch.randn(N, C, H, W, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gn = torch.nn.GroupNorm(G, C, eps=1e-5).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, device="cuda", generator=g) * 0.5 + 1.0)
        gn.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.2)
    xr = x.float().requires_grad_(True)
    ref = F.group_norm(xr, G, gn.weight, gn.bias, gn.eps)
    if silu:
        ref = ref * torch.sigmoid(ref)
    xg = x.clone().requires_grad_(True)
    gn.weight.grad = gn.bias.grad = None
    y = norm.fused_gn_act(xg, gn, silu=silu)
    assert y.dtype == torch.bfloat16 and y.shape == ref.shape

    def close(got, want, what):
        got, want = got.float(), want.float()
        bad = (got - want).abs() > want.abs() * 2.0 ** -8 + 1e-5 * want.abs().max() + 1e-6
        assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside, worst {(got - want).abs().max():.3e}"

    close(y, ref, "forward")
    dy = torch.randn(ref.shape, device="cuda", generator=g).to(torch.bfloat16)
    gw_ref, gb_ref, gx_ref = torch.autograd.grad(ref, [gn.weight, gn.bias, xr], dy.float())
    y.backward(dy)
    close(xg.grad, gx_ref, "dx")
    assert (gn.weight.grad - gw_ref).abs().max() <= 1e-4 * gw_ref.abs().max() + 1e-5
    assert (gn.bias.grad - gb_ref).abs().max() <= 1e-4 * gb_ref.abs().max() + 1e-5
    # a second backward accumulates into the existing .grad (the flat-arena path)
    y2 = norm.fused_gn_act(xg, gn, silu=silu)
    y2.backward(dy)
    assert (gn.weight.grad - 2 * gw_ref).abs().max() <= 2e-4 * gw_ref.abs().max() + 2e-5
