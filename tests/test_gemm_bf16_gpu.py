"""K16 (csrc/salun_gemm.hip::k_gemm_bf16_nt): the bf16 direct-to-LDS GEMM behind the SD transformer blocks' Linear layers
(reference: autocast over SD/ldm/modules/attention.py:37-75,149-200), through the C-ABI, against an fp32 matmul of the
SAME bf16-rounded operands (what differs is the order of fp32 accumulation and the final rounding: <= 1 bf16 ulp of the
result's scale), for every tile variant, ragged token counts and the fused bias / residual epilogue."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from unlearn_saliency_amd import rng

pytestmark = pytest.mark.gpu


def dev(shape, seed, std=1.0):
    return torch.from_numpy(rng.normal(int(np.prod(shape)), seed, 0.0, std)).view(*shape).cuda()


def rel(got, want):
    return float((got.double() - want.double()).abs().max() / want.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 640, 320), (616, 1280, 768), (8192, 2560, 320), (77, 128, 1280)])
def test_gemm_bf16_nt_matches_fp32_matmul_of_the_rounded_operands(variant, M, N, K):
    from unlearn_saliency_amd import ops
    x, w = dev((M, K), 1).bfloat16(), dev((N, K), 2, 0.05).bfloat16()
    bias, add = dev((N,), 3, 0.1), dev((M, N), 4).bfloat16()
    want = x.float() @ w.float().t()
    for b, a in ((None, None), (bias, None), (bias, add)):
        ref = want + (b if b is not None else 0) + (a.float() if a is not None else 0)
        got = ops.gemm_bf16_nt(x, w, b, a, variant)
        assert got.dtype == torch.bfloat16 and got.shape == (M, N)
        e = rel(got, ref)
        assert e <= 6e-3, (variant, M, N, K, b is not None, a is not None, e)  # bf16 rounding of the result: 2^-8
        # the exact check: round the fp32 reference the same way — at most one bf16 ulp apart anywhere
        ulp = (ref.abs().clamp_min(1e-3) * 2.0 ** -7)
        assert bool(((got.float() - ref).abs() <= ulp).all())
    assert torch.equal(ops.gemm_bf16_nt(x, w, bias, add, variant), ops.gemm_bf16_nt(x, w, bias, add, variant))


def test_variants_agree_bitwise_and_shapes_outside_the_domain_are_refused():
    from unlearn_saliency_amd import _lib, ops
    x, w = dev((4096, 320), 5).bfloat16(), dev((640, 320), 6, 0.05).bfloat16()
    outs = [ops.gemm_bf16_nt(x, w, None, None, v) for v in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])  # same k order in every tile shape: identical fp32 sums
    assert not ops.gemm_bf16_supported(128, 96, 64) and not ops.gemm_bf16_supported(128, 64, 40)
    with pytest.raises(_lib.SalunError):
        ops.gemm_bf16_nt(x[:, :40].contiguous(), w[:, :40].contiguous())
    with pytest.raises(_lib.SalunError):
        ops.gemm_bf16_nt(x, w[:320].contiguous(), variant=1)  # N = 320 is not a multiple of the 128-feature tile


def test_pack_bf16_both_images():
    from unlearn_saliency_amd import ops
    w = dev((320, 1280), 7, 0.05)
    assert torch.equal(ops.pack_bf16(w), w.bfloat16())
    assert torch.equal(ops.pack_bf16(w, transposed=True), w.t().contiguous().bfloat16())
    odd = dev((37, 53), 8)
    assert torch.equal(ops.pack_bf16(odd, transposed=True), odd.t().contiguous().bfloat16())


@pytest.mark.parametrize("M", [4096, 616])
def test_linear_bf16_on_the_gemm_matches_the_fp32_layer(M):
    """SalunLinearBF16 (forward and input gradient on K16, weight gradient on K11) vs the fp32 nn.Linear on the
    bf16-rounded input: outputs within bf16 rounding, gradients within the bf16 tolerance of this configuration."""
    from unlearn_saliency_amd import conv_bf16
    from unlearn_saliency_amd.flat import arena_of
    torch.manual_seed(0)
    lin = torch.nn.Linear(320, 1280).cuda()
    ref = torch.nn.Linear(320, 1280).cuda()
    ref.load_state_dict(lin.state_dict())
    lin.__class__ = conv_bf16.SalunLinearBF16
    arena_of(lin)
    x = dev((M, 320), 11).bfloat16().requires_grad_(True)
    res = dev((M, 1280), 12).bfloat16()
    dy = dev((M, 1280), 13).bfloat16()
    y = lin(x, addend=res)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    yr = ref(xr) + res.float()
    yr.backward(dy.float())
    assert rel(y, yr) <= 6e-3
    assert rel(x.grad, xr.grad) <= 1e-2
    assert rel(lin.weight.grad, ref.weight.grad) <= 1e-2 and rel(lin.bias.grad, ref.bias.grad) <= 1e-2
    # K16 vs the K11 1x1 route behind the same module: same operands, fp32 accumulation in both
    conv_bf16._USE_K16[0] = False
    try:
        x2 = x.detach().clone().requires_grad_(True)
        lin.zero_grad(set_to_none=False)
        y2 = lin(x2, addend=res)
        y2.backward(dy)
    finally:
        conv_bf16._USE_K16[0] = True
    assert rel(y2, y) <= 8e-3 and rel(x2.grad, x.grad) <= 8e-3


def test_transformer_block_with_residual_epilogues_matches_the_plain_block():
    """BasicTransformerBlock in the bf16 configuration: Linear layers on K16 with the three residual adds folded into
    the output projections' epilogues vs the same block with library Linear layers (autocast)."""
    import copy
    from fixtures import fill_params
    from unlearn_saliency_amd.conv_bf16 import use_salun_linears_bf16
    from unlearn_saliency_amd.SD.unet import BasicTransformerBlock
    blk = fill_params(BasicTransformerBlock(320, 8, 40, context_dim=768, use_checkpoint=False), 100).cuda()
    plain = copy.deepcopy(blk)
    assert use_salun_linears_bf16(torch.nn.ModuleList([blk])) >= 8
    x = dev((2, 1024, 320), 21).bfloat16()
    ctx = dev((2, 77, 768), 22)
    outs = []
    for m in (blk, plain):
        xx = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xx, ctx)
        y.float().backward(dev((2, 1024, 320), 23))
        outs.append((y.detach().float(), xx.grad.float(), {n: p.grad.float() for n, p in m.named_parameters()}))
    (y1, dx1, g1), (y0, dx0, g0) = outs
    assert rel(y1, y0) <= 3e-2 and rel(dx1, dx0) <= 3e-2
    for n in g0:
        assert rel(g1[n], g0[n]) <= 5e-2, n


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("M,Na,Nb", [(64, 32, 32), (616, 320, 768), (4096, 1280, 320), (1000, 96, 160), (2048, 1280, 1280),
                                     (33, 64, 64), (32768, 320, 320)])
def test_gemm_bf16_tn_weight_gradient_matches_fp32_matmul_of_the_rounded_operands(variant, M, Na, Nb):
    """dW = dY^T . X (the reduction over tokens is the slow axis of both operands: transposing LDS reads), ragged token
    counts (tokens past M read zeros), feature counts that end inside a 128-wide tile, split reductions, accumulation."""
    from unlearn_saliency_amd import ops
    dy, x = dev((M, Na), 11).bfloat16(), dev((M, Nb), 12).bfloat16()
    want = dy.double().t() @ x.double()
    got = ops.gemm_bf16_tn(dy, x, variant=variant)
    assert got.dtype == torch.float32 and got.shape == (Na, Nb)
    e = rel(got, want)
    assert e <= 2e-6 * max(1.0, (M / 4096) ** 0.5), (variant, M, Na, Nb, e)   # fp32 accumulation of exact bf16 products
    assert torch.equal(ops.gemm_bf16_tn(dy, x, variant=variant), got)  # fixed split order, no atomics
    base = dev((Na, Nb), 13)
    acc = base.clone()
    ops.gemm_bf16_tn(dy, x, out=acc, accumulate=True, variant=variant)
    assert rel(acc, base.double() + want) <= 3e-6 * max(1.0, (M / 4096) ** 0.5)
    # transpose-detecting: the operands are not symmetric, so dW^T would fail the first check; also X and dY swapped
    if Na == Nb:
        assert rel(got.t(), want) > 1e-2


def test_conv_1x1_and_linear_weight_gradients_take_the_tn_kernel_and_agree_with_the_tap_kernel():
    """salun_conv2d_bf16_backward_weight routes 1x1 / stride 1 / no padding to the TN GEMM; the bias gradient rides along."""
    from unlearn_saliency_amd import ops
    N, H, W, C, K = 2, 16, 16, 320, 640
    x, dy = dev((N, H, W, C), 21).bfloat16(), dev((N, H, W, K), 22).bfloat16()
    db = torch.zeros(K, device="cuda")
    dw = ops.conv2d_bf16_backward_weight(x, dy, (K, C, 1, 1), 1, 0, bias_out=db)
    want = dy.view(-1, K).double().t() @ x.view(-1, C).double()
    assert rel(dw.view(K, C), want) <= 2e-6
    assert rel(db, dy.view(-1, K).double().sum(0)) <= 2e-6
    # accumulate into an existing gradient (the gradient sinks of the flat arena)
    dw2 = dw.clone()
    ops.conv2d_bf16_backward_weight(x, dy, (K, C, 1, 1), 1, 0, out=dw2, accumulate=True)
    assert rel(dw2, 2 * dw.double()) <= 2e-6
