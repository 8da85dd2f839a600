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


def test_batched_weight_images_equal_the_single_launch_images_bit_for_bit():
    """salun_bf16_pack_weights_batch (one launch per 64 layers, round 6) writes what salun_conv2d_bf16_pack_weights /
    salun_pack_bf16(transposed) write — 3x3, 1x1, Linear, transposed Linear, channel counts that are not multiples of 32
    in the transposed tiles, more jobs than one launch takes."""
    from unlearn_saliency_amd import ops
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 32, 3), (320, 640, 3), (96, 64, 1), (1280, 320, 1), (32, 32, 3)] + [(32 + 8 * (i % 5), 40 + 8 * (i % 3), 1) for i in range(70)]
    jobs, want = [], []
    for K, C, R in shapes:
        w = torch.randn(K, C, R, R, generator=g).cuda()
        img = torch.empty((K, R * R, C), dtype=torch.bfloat16, device="cuda")
        jobs.append((w, img, K, C, R, False))
        want.append(ops.conv2d_bf16_pack(w))
        if R == 1:
            imt = torch.empty((C, K), dtype=torch.bfloat16, device="cuda")
            jobs.append((w.view(K, C), imt, K, C, 1, True))
            want.append(ops.pack_bf16(w.view(K, C), True))
    n = ops.bf16_pack_batch(jobs)
    assert n == (len(jobs) + 63) // 64 and n >= 2
    for (w, img, *_), ref in zip(jobs, want):
        assert torch.equal(img.view(torch.int16), ref.view(torch.int16).view(img.shape))


def test_a_model_packs_all_its_layers_in_one_batch_per_step():
    from unlearn_saliency_amd import conv_bf16, ops
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(32, 64, 3, padding=1), torch.nn.Conv2d(64, 64, 1), torch.nn.Conv2d(64, 32, 3, padding=1)).cuda()
    assert conv_bf16.use_salun_convs_bf16(net) == 3
    x = torch.randn(2, 32, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = lambda: torch.nn.Sequential(*[torch.nn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, padding=m.padding) for m in net])
    n0 = conv_bf16.PACK_LAUNCHES[0]
    y = net(x)
    assert conv_bf16.PACK_LAUNCHES[0] == n0 + 1
    conv_bf16.BATCH_PACKS[0] = False
    try:
        ops.PARAM_EPOCH[0] += 1
        y1 = net(x)                      # one launch per layer: the same images
        assert conv_bf16.PACK_LAUNCHES[0] == n0 + 4 and torch.equal(y, y1)
    finally:
        conv_bf16.BATCH_PACKS[0] = True
    with torch.no_grad():
        net[1].weight.mul_(0.5)
    y2 = net(x)
    assert conv_bf16.PACK_LAUNCHES[0] == n0 + 5 and not torch.equal(y, y2)


# (N, H, W, C, K, R): the tap kernels (3x3), the dY^T.X GEMM route (1x1 with >= 1024 pixels), images smaller than one chunk,
# a batch that does not divide the 128 chunk slots, the 1x1 tap route (few pixels)
@pytest.mark.parametrize("dims", [(8, 16, 16, 64, 96, 3), (8, 32, 32, 64, 128, 1), (3, 5, 7, 32, 160, 3), (48, 8, 8, 32, 64, 3),
                                  (2, 8, 8, 64, 64, 1), (128, 4, 4, 32, 32, 3)])
def test_per_image_channel_sums_ride_on_the_bias_gradient(dims):
    """salun_conv2d_bf16_backward_weight_ex: dnb[n][k] = sum over the pixels of image n of dy (the gradient of the forward's
    nbias term) from the bias gradient's partial sums; db and dw are what the call without dnb gives (db: another
    summation order, fp32 rounding)."""
    from unlearn_saliency_amd import ops
    N, H, W, C, K, R = dims
    x, dy = _mk((N, H, W, C), 11), _mk((N, H, W, K), 12)
    pad = R // 2
    db0 = torch.zeros(K, device="cuda")
    dw0 = ops.conv2d_bf16_backward_weight(x, dy, (K, C, R, R), 1, pad, bias_out=db0)
    db1 = torch.zeros(K, device="cuda")
    dnb = torch.full((N, K), float("nan"), device="cuda")
    dw1 = ops.conv2d_bf16_backward_weight(x, dy, (K, C, R, R), 1, pad, bias_out=db1, nbias_out=dnb)
    assert torch.equal(dw0, dw1)
    ref = dy.double().sum(dim=(1, 2))
    scale = float(ref.abs().max())
    assert float((dnb.double() - ref).abs().max()) <= 1e-5 * scale
    assert float((db1.double() - ref.sum(0)).abs().max()) <= 1e-5 * float(ref.sum(0).abs().max() + scale)
    assert float((db0 - db1).abs().max()) <= 1e-5 * float(db0.abs().max())
    # without a bias gradient, and accumulating db: dnb is overwritten all the same
    dnb2 = torch.full((N, K), 7.0, device="cuda")
    acc = torch.ones(K, device="cuda")
    out = torch.zeros(K, C, R, R, device="cuda")
    ops.conv2d_bf16_backward_weight(x, dy, (K, C, R, R), 1, pad, out=out, accumulate=True, bias_out=acc, nbias_out=dnb2)
    assert torch.equal(dnb2, dnb) and float((acc - 1 - db1).abs().max()) <= 1e-5 * float(db1.abs().max())
    dnb3 = torch.empty((N, K), device="cuda")
    ops.conv2d_bf16_backward_weight(x, dy, (K, C, R, R), 1, pad, nbias_out=dnb3)
    assert torch.equal(dnb3, dnb)
    # the sums alone (salun_colsum_bf16): the same kernels, the same numbers
    dnb4, db4 = torch.empty((N, K), device="cuda"), torch.ones(K, device="cuda")
    ops.colsum_bf16(dy, N, bias_out=db4, nbias_out=dnb4, accumulate=True)
    assert torch.equal(dnb4, dnb) and torch.equal(db4, acc)
    db5 = torch.empty(K, device="cuda")
    ops.colsum_bf16(dy, N, bias_out=db5)
    assert torch.equal(db5, db0)


def test_resblock_time_embedding_gradient_through_the_fused_sums():
    """SalunConv2dBF16(x, nbias=emb): emb.grad from the kernel equals dy.float().sum over the pixels (what autograd computed
    from a copy of dy before round 6), weights trainable or frozen."""
    from unlearn_saliency_amd import conv_bf16
    torch.manual_seed(3)
    for frozen, sunk in ((False, False), (True, False), (False, True)):
        conv = torch.nn.Conv2d(64, 96, 3, padding=1).cuda()
        conv.__class__ = conv_bf16.SalunConv2dBF16
        conv.weight.requires_grad_(not frozen)
        conv.bias.requires_grad_(not frozen)
        if sunk:  # gradients go straight into .grad: backward-weight on the side stream, the sums on this one
            conv.weight.grad, conv.bias.grad = torch.zeros_like(conv.weight), torch.zeros_like(conv.bias)
        x = torch.randn(4, 64, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        emb = torch.randn(4, 96, device="cuda", requires_grad=True)
        y = conv(x, nbias=emb)
        dy = torch.randn_like(y)
        y.backward(dy)
        ref = dy.float().sum(dim=(2, 3))
        assert float((emb.grad - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        if sunk:
            torch.cuda.synchronize()
            assert float((conv.bias.grad - ref.sum(0)).abs().max()) <= 1e-5 * float(ref.sum(0).abs().max() + ref.abs().max())
            assert float(conv.weight.grad.abs().max()) > 0
