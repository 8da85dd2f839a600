"""K11 (csrc/salun_conv_bf16.hip): bf16 NHWC convolution forward / backward-data / backward-weight on the bf16 MFMA
instruction against the library fp32 convolution evaluated on the SAME bf16-rounded inputs.

Tolerance: products of bf16 values are exact in fp32 and both sides accumulate in fp32, so the two differ by
summation order (~1e-6 relative to the reduction's magnitude) plus, for the bf16 outputs (forward, backward-data), the
final rounding to bf16 (relative 2^-9).  Asserted: |got - ref| <= 2^-8 * |ref| + 1e-5 * max|ref| for bf16 outputs,
<= 2e-5 * max|ref| for the fp32 weight gradient.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (N, H, W, C, K, R, stride, pad)
SHAPES = [
    (2, 8, 8, 32, 32, 3, 1, 1),        # the tiny U-Net's level
    (2, 8, 8, 64, 32, 1, 1, 0),
    (1, 5, 7, 32, 96, 3, 1, 1),        # ragged pixel / channel tiles
    (2, 16, 16, 64, 128, 3, 1, 1),
    (2, 16, 16, 96, 64, 3, 2, 1),      # stride 2 (Downsample), C % 64 != 0
    (1, 9, 9, 64, 64, 3, 2, 1),        # odd extent with stride 2
    (8, 32, 32, 320, 320, 3, 1, 1),    # SD level 0/1 shape class (reduced extent)
    (8, 16, 16, 640, 1280, 3, 1, 1),
    (8, 16, 16, 1280, 640, 1, 1, 0),
    (8, 16, 16, 640, 640, 3, 2, 1),
    (2, 64, 64, 320, 320, 3, 1, 1),    # full 64x64 latent extent
    (8, 8, 8, 2560, 1280, 3, 1, 1),    # decoder entry (concatenated skip)
    # shapes with >= 128 forward tiles: the 1x1 one takes the LDS-DMA ring kernel (k_conv_bf16_ring, salun_gemm.hip) by
    # default, all four do under SALUN_CONV_RING=3 (test_ring_kernel_every_tile_form_in_a_child_process)
    (4, 64, 64, 32, 256, 3, 1, 1),     # 256 x 128 tiles
    (4, 64, 64, 64, 256, 1, 1, 0),     # 256 x 128, 1x1
    (5, 62, 62, 64, 192, 3, 1, 1),     # 256 x 64 tiles, ragged last pixel tile
    (4, 128, 128, 32, 128, 3, 2, 1),   # 128 x 128 tiles, stride 2
]


def _mk(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(torch.bfloat16)


def _close_bf16(got, ref, what):
    got, ref = got.float(), ref.float()
    bound = ref.abs() * 2.0 ** -8 + 1e-5 * ref.abs().max()
    bad = (got - ref).abs() > bound
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} outside the bound, worst {(got - ref).abs().max():.3e}"


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_backward_match_library_on_bf16_inputs(shape):
    from unlearn_saliency_amd import ops
    N, H, W, C, K, R, st, pad = shape
    assert ops.conv2d_bf16_supported(C, K, R, st, pad)
    x = _mk((N, H, W, C), 1)
    w = torch.randn(K, C, R, R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)) / (C * R * R) ** 0.5
    bias = torch.randn(K, device="cuda")
    wp = ops.conv2d_bf16_pack(w)
    w_r = w.to(torch.bfloat16).float()
    assert torch.equal(wp.float(), w_r.permute(0, 2, 3, 1).reshape(K, R * R, C))
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wr = w_r.clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, wr, bias, st, pad)
    y = ops.conv2d_bf16_forward(x, wp, R, st, pad, bias=bias)
    assert y.shape == (N, ref.shape[2], ref.shape[3], K)
    _close_bf16(y, ref.permute(0, 2, 3, 1), "forward")
    dy = _mk(tuple(y.shape), 3)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dx = ops.conv2d_bf16_backward_data(dy, wp, (N, H, W, C), R, st, pad)
    _close_bf16(dx, xr.grad.permute(0, 2, 3, 1), "backward-data")
    db = torch.zeros(K, device="cuda")
    dw = ops.conv2d_bf16_backward_weight(x, dy, (K, C, R, R), st, pad, bias_out=db)
    assert (dw - wr.grad).abs().max() <= 2e-5 * wr.grad.abs().max(), (dw - wr.grad).abs().max() / wr.grad.abs().max()
    ref_db = dy.float().sum((0, 1, 2))
    assert (db - ref_db).abs().max() <= 1e-5 * ref_db.abs().max() + 1e-4
    # accumulate into an existing gradient
    acc = torch.ones_like(dw)
    ops.conv2d_bf16_backward_weight(x, dy, (K, C, R, R), st, pad, out=acc, accumulate=True)
    assert (acc - 1 - dw).abs().max() <= 1e-6 * max(1.0, float(dw.abs().max()))


# (N, H, W, C, K): 64 channels -> the reduction is split and k_splitk_finish carries the terms; 32 channels -> nine stages,
# no split: conv_bf16_igemm's own epilogue (ragged pixel and channel tiles in the second case)
@pytest.mark.parametrize("dims", [(2, 8, 8, 64, 64), (2, 8, 8, 32, 32), (1, 5, 7, 32, 96), (3, 16, 16, 32, 160)])
def test_epilogue_terms(dims):
    from unlearn_saliency_amd import ops
    N, H, W, C, K = dims
    x, w = _mk((N, H, W, C), 5), torch.randn(K, C, 3, 3, device="cuda") / (3.0 * C ** 0.5)
    wp = ops.conv2d_bf16_pack(w)
    nb = torch.randn(N, K, device="cuda")
    add = _mk((N, H, W, K), 6)
    base = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), None, 1, 1).permute(0, 2, 3, 1)
    y = ops.conv2d_bf16_forward(x, wp, 3, 1, 1, nbias=nb, addend=add)
    _close_bf16(y, base + nb[:, None, None, :] + add.float(), "forward + nbias + addend")
    dy = _mk((N, H, W, K), 7)
    dx_ref = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), None, 1, 1)
    addx = _mk((N, H, W, C), 8)
    dx1 = ops.conv2d_bf16_backward_data(dy, wp, (N, H, W, C), 3, 1, 1, addend=addx)
    _close_bf16(dx1, dx_ref.permute(0, 2, 3, 1) + addx.float(), "backward-data + addend")


def test_ring_kernel_every_tile_form_in_a_child_process():
    """SALUN_CONV_RING=3 sends every eligible forward launch (>= 128 tiles) to the LDS-DMA ring kernel (by default only the
    1x1 convolutions go there: the 3x3 ones are faster on the register-staged kernel); the switch is read once per process,
    so the 3x3 / stride-2 / ragged shapes of SHAPES and the epilogue terms are re-run in a child."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, torch; sys.path.insert(0, %r); import test_conv_bf16_gpu as t\n"
            "from unlearn_saliency_amd import ops\n"
            "for sh in t.SHAPES[-4:] + [(8, 32, 32, 320, 320, 3, 1, 1), (8, 16, 16, 640, 1280, 3, 1, 1)]:\n"
            "    t.test_forward_backward_match_library_on_bf16_inputs(sh)\n"
            "N,H,W,C,K=4,64,64,64,256; x=t._mk((N,H,W,C),5); w=torch.randn(K,C,3,3,device='cuda')/24.0\n"
            "wp=ops.conv2d_bf16_pack(w); bias=torch.randn(K,device='cuda'); nb=torch.randn(N,K,device='cuda'); add=t._mk((N,H,W,K),6)\n"
            "base=torch.nn.functional.conv2d(x.float().permute(0,3,1,2), w.to(torch.bfloat16).float(), bias, 1, 1).permute(0,2,3,1)\n"
            "y=ops.conv2d_bf16_forward(x,wp,3,1,1,bias=bias,nbias=nb,addend=add)\n"
            "t._close_bf16(y, base + nb[:,None,None,:] + add.float(), 'ring forward + bias + nbias + addend')\n"
            "print('ring ok')") % here
    env = dict(os.environ, SALUN_CONV_RING="3", PYTHONPATH=os.path.dirname(here))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ring ok"), out.stderr[-3000:]


def test_unsupported_shapes_are_refused():
    from unlearn_saliency_amd import ops
    assert not ops.conv2d_bf16_supported(4, 320, 3, 1, 1)      # the RGB-like stem stays on the fp32 kernels
    assert not ops.conv2d_bf16_supported(320, 4, 3, 1, 1)
    assert not ops.conv2d_bf16_supported(64, 64, 5, 1, 2)


@pytest.mark.parametrize("M_shape,C,K,bias", [((8, 4096), 320, 320, False), ((8, 77), 768, 320, False),
                                              ((2, 1024), 640, 5120, True), ((3, 50), 64, 96, True)])
def test_linear_on_the_1x1_kernels_matches_fp32_linear(M_shape, C, K, bias):
    """SalunLinearBF16 (a Linear layer = 1x1 convolution over the tokens, K11 kernels) vs F.linear in fp32 on the same
    bf16-rounded inputs: forward, dx, dW, db, and the residual in the epilogue.  Tolerance 2e-2 of each tensor's scale
    (bf16 output / input-gradient rounding; dW / db are fp32 sums of bf16 products: 1e-2)."""
    from unlearn_saliency_amd.conv_bf16 import SalunLinearBF16
    g = torch.Generator(device="cuda").manual_seed(sum(M_shape) + C)
    lin = torch.nn.Linear(C, K, bias=bias).cuda()
    lin.__class__ = SalunLinearBF16
    x = torch.randn(*M_shape, C, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    res = torch.randn(*M_shape, K, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(*M_shape, K, device="cuda", generator=g).to(torch.bfloat16)
    y = lin(x, addend=res)
    assert y.dtype == torch.bfloat16 and y.shape == (*M_shape, K)
    y.backward(dy)
    wq = lin.weight.detach().to(torch.bfloat16).float()
    xf = x.detach().float().requires_grad_(True)
    wf = wq.clone().requires_grad_(True)
    bf = lin.bias.detach().clone().requires_grad_(True) if bias else None
    yr = torch.nn.functional.linear(xf, wf, bf) + res.detach().float()
    yr.backward(dy.float())
    rel = lambda a, b: float((a.float() - b).abs().max() / b.abs().max().clamp_min(1e-6))
    errs = {"y": rel(y, yr), "dx": rel(x.grad, xf.grad), "dw": rel(lin.weight.grad, wf.grad),
            "dres": rel(res.grad, dy.float())}
    if bias:
        errs["db"] = rel(lin.bias.grad, bf.grad)
    print(M_shape, C, K, {k: f"{v:.1e}" for k, v in errs.items()})
    assert errs["y"] <= 2e-2 and errs["dx"] <= 2e-2 and errs["dw"] <= 1e-2 and errs["dres"] == 0.0
    assert not bias or errs["db"] <= 1e-2
