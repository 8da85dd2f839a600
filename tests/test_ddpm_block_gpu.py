"""The diffusion ResnetBlock as one autograd node (resblock.py::_DiffusionResnetBlockFn; reference block:
DDPM/models/diffusion.py:85-128) and the two kernel epilogues it is made of:

* `salun_conv2d_forward_fused`   y = conv(x, w) + bias[k] + nbias[n][k] + addend          vs float64
* `salun_gn_backward_fused`      dx (+= addend), nk = sum_hw dx, csum = sum_n nk          vs the plain kernel + sums
* the node                       output, dx, dproj and every parameter gradient           vs the same block as
                                 separate autograd nodes (identity / 1x1 / 3x3 skip; with dropout under one key)
* the whole reduced U-Net        loss and all 100+ parameter gradients, nodes on vs off, gradients in the flat arena

Tolerances are relative to each tensor's scale; the only arithmetic that differs between the two routes is the order of
the bias / projection gradient sums, so 1e-5 is generous (the measured worst is printed)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fixtures import ddpm_batch, ddpm_small_config, fill_params
from unlearn_saliency_amd import rng

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev_normal(shape, seed, std=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy(rng.normal(n, seed, 0.0, std)).view(*shape).cuda()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


# (128, 256, 256, 3, 4) and (128, 512, 256, 3, 4): the 4x4 level of the DDPM U-Net at batch 128 — 128 workgroups, run
# with the reduction split (four workgroups per output tile + the finishing kernel carrying the epilogue terms)
@pytest.mark.parametrize("shape", [(4, 64, 96, 3, 16), (3, 128, 128, 3, 8), (4, 96, 64, 1, 16), (130, 128, 128, 3, 4),
                                   (128, 256, 256, 3, 4), (128, 512, 256, 3, 4), (128, 256, 256, 1, 4)])
def test_forward_epilogue_bias_nbias_addend(shape):
    from unlearn_saliency_amd import ops
    N, C, K, R, H = shape
    pad = (R - 1) // 2
    x, w, b = dev_normal((N, C, H, H), 1), dev_normal((K, C, R, R), 2, 0.05), dev_normal((K,), 3)
    nb, ad = dev_normal((N, K), 4), dev_normal((N, K, H, H), 5)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, pad)
    for use_b, use_nb, use_ad in [(1, 1, 1), (0, 1, 0), (1, 0, 1), (0, 0, 1), (1, 0, 0)]:
        y = ops.conv2d_forward(x, w, b if use_b else None, 1, pad, H, H, nbias=nb if use_nb else None,
                               addend=ad if use_ad else None)
        assert y is not None
        want = ref - (0 if use_b else b.double()[None, :, None, None])
        if use_nb:
            want = want + nb.double()[:, :, None, None]
        if use_ad:
            want = want + ad.double()
        e = rel(y, want)
        assert e <= TOL, (shape, use_b, use_nb, use_ad, e)
    # the plain entry point is the fused one with nothing attached: bit-identical
    y0 = ops.conv2d_forward(x, w, b, 1, pad, H, H)
    with torch.no_grad():
        y1 = ops.conv2d_forward(x, w, b, 1, pad, H, H, nbias=torch.zeros_like(nb), addend=torch.zeros_like(ad))
    assert torch.equal(y0, y1)
    with pytest.raises(ValueError):
        ops.conv2d_forward(x, w, b, 1, pad, H, H, nbias=nb[:, :-1].contiguous())


@pytest.mark.parametrize("shape", [(4, 128, 16, 16), (3, 256, 8, 8), (2, 64, 32, 32), (2, 320, 64, 64), (5, 96, 2, 2)])
@pytest.mark.parametrize("silu", [True, False])
def test_gn_backward_fused_outputs(shape, silu):
    from unlearn_saliency_amd import ops
    N, C, H, W = shape
    x, dz, ad = dev_normal(shape, 11), dev_normal(shape, 12), dev_normal(shape, 13)
    g, b = 1.0 + 0.1 * dev_normal((C,), 14), 0.1 * dev_normal((C,), 15)
    z, m, r = ops.gn_forward(x, g, b, 32, 1e-6, silu)
    dx0, dg0, db0 = ops.gn_backward(dz, x, g, b, m, r, 32, silu)
    # against autograd in float64
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yd = F.group_norm(xd, 32, gd, bd, 1e-6)
    if silu:
        yd = yd * torch.sigmoid(yd)
    yd.backward(dz.double())
    assert rel(dx0, xd.grad) <= TOL and rel(dg0, gd.grad) <= TOL and rel(db0, bd.grad) <= TOL
    acc = torch.full((C,), 0.5, device="cuda")
    dx1, dg1, db1, nk, cs = ops.gn_backward(dz, x, g, b, m, r, 32, silu, addend=ad, nk_sum=True, csum=True,
                                            csum_acc=acc)
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0)     # same reductions, same order
    assert torch.equal(dx1, dx0 + ad)                           # one rounding: fl(dx + addend), as the separate add
    want_nk = dx1.double().sum(dim=(2, 3))
    scale = float(dx1.double().abs().sum(dim=(2, 3)).max())
    assert float((nk.double() - want_nk).abs().max()) <= 1e-6 * scale
    assert float((cs.double() - want_nk.sum(0)).abs().max()) <= 1e-6 * scale * N
    assert torch.allclose(acc, 0.5 + cs, rtol=0, atol=1e-6 * scale * N)
    # nk without addend
    _, _, _, nk2, cs2 = ops.gn_backward(dz, x, g, b, m, r, 32, silu, nk_sum=True)
    assert cs2 is None
    assert float((nk2.double() - dx0.double().sum(dim=(2, 3))).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("shape", [(128, 256, 256, 3, 4), (128, 512, 256, 3, 4), (128, 256, 256, 1, 4), (16, 128, 128, 3, 8)])
def test_reduction_split_backward_data_and_workspace(shape):
    """Stride-1 backward-data of an under-filled launch (split over the reduction channels) against float64, with and
    without the addend (also in place: addend is dx), and the workspace query that announces the split."""
    from unlearn_saliency_amd import _lib, ops
    N, C, K, R, H = shape
    pad = (R - 1) // 2
    L = _lib.lib()
    assert L.salun_conv2d_data_workspace_bytes(N, C, H, H, R, 1) == 8 * N * C * H * H * 4
    assert L.salun_conv2d_data_workspace_bytes(256, 64, 32, 32, 3, 1) == 0      # a full launch is never split
    w, dy, ad = dev_normal((K, C, R, R), 41, 0.05), dev_normal((N, K, H, H), 42), dev_normal((N, C, H, H), 43)
    want = torch.nn.grad.conv2d_input((N, C, H, H), w.double(), dy.double(), 1, pad)
    dx = ops.conv2d_backward_data(dy, w, (N, C, H, H), 1, pad)
    assert rel(dx, want) <= TOL
    dxa = ops.conv2d_backward_data(dy, w, (N, C, H, H), 1, pad, addend=ad)
    assert rel(dxa, want + ad.double()) <= TOL and torch.equal(dxa, dx + ad)
    # the unsplit launch (no workspace through the legacy entry point) agrees to rounding
    from ctypes import c_void_p
    dx0 = torch.empty_like(dx)
    assert L.salun_conv2d_backward_data(c_void_p(dy.data_ptr()), c_void_p(w.data_ptr()), c_void_p(dx0.data_ptr()), N, C, H,
                                        H, K, R, 1, pad, H, H, c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    assert rel(dx0, want) <= TOL and rel(dx, dx0) <= TOL


def _block_pair(cin, cout, conv_shortcut, p):
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.DDPM.models.diffusion import ResnetBlock
    blk = fill_params(ResnetBlock(in_channels=cin, out_channels=cout, conv_shortcut=conv_shortcut, dropout=p), 500).cuda()
    sep = copy.deepcopy(blk)
    assert use_salun_convs(blk) > 0 and use_salun_convs(sep) > 0
    assert blk.fused_node
    sep.fused_node = False
    return blk.train(), sep.train()


@pytest.mark.parametrize("cin,cout,conv_shortcut,p", [(128, 128, False, 0.0), (128, 256, False, 0.0),
                                                      (256, 128, True, 0.0), (256, 256, False, 0.1)])
def test_node_matches_separate_nodes(monkeypatch, cin, cout, conv_shortcut, p):
    from unlearn_saliency_amd import conv as sconv
    blk, sep = _block_pair(cin, cout, conv_shortcut, p)
    N, H = 4, 16
    x0, emb, dout = dev_normal((N, cin, H, H), 21), dev_normal((N, 1024), 22), dev_normal((N, cout, H, H), 23)
    from unlearn_saliency_amd import draws
    outs = []
    sconv.reset_library_conv_calls()
    for m in (blk, sep):
        draws.set_state((1234, 7, 0))  # both routes take the same dropout key: the keep mask is a function of it alone
        x = x0.clone().requires_grad_(True)
        e = emb.clone().requires_grad_(True)
        y = m(x, e)
        y.backward(dout)
        torch.cuda.synchronize()
        outs.append((y.detach(), x.grad, e.grad, {n: q.grad for n, q in m.named_parameters()}))
    assert sconv.library_conv_calls() == 0
    (y1, dx1, de1, g1), (y2, dx2, de2, g2) = outs
    worst = max(rel(y1, y2), rel(dx1, dx2), rel(de1, de2))
    assert worst <= TOL, worst
    assert set(g1) == set(g2) and all(v is not None for v in g1.values())
    ptrs = [v.data_ptr() for v in g1.values()]
    assert len(set(ptrs)) == len(ptrs)  # no two parameters share one gradient tensor (conv2.bias / skip bias)
    for n in g1:
        e = rel(g1[n], g2[n])
        worst = max(worst, e)
        assert e <= TOL, (n, e)
    print(f"ResnetBlock {cin}->{cout} skip3x3={conv_shortcut} p={p}: node vs separate nodes, worst {worst:.2e} of scale")


def test_node_falls_back_outside_its_domain():
    """CPU tensors, autocast and the fused-GroupNorm switch send the block down its ordinary forward."""
    from unlearn_saliency_amd import norm
    from unlearn_saliency_amd.resblock import fused_diffusion_resnet_block
    blk, _ = _block_pair(128, 128, False, 0.0)
    x, emb = dev_normal((2, 128, 8, 8), 31), dev_normal((2, 1024), 32)
    assert fused_diffusion_resnet_block(blk, x, emb) is not None
    assert fused_diffusion_resnet_block(blk, x.half(), emb) is None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert fused_diffusion_resnet_block(blk, x, emb) is None
    norm.enable_fused_gn(False)
    try:
        assert fused_diffusion_resnet_block(blk, x, emb) is None
    finally:
        norm.enable_fused_gn(True)
    odd = dev_normal((2, 128, 6, 6), 33)   # H*W not a power of two: GroupNorm kernel's domain
    assert fused_diffusion_resnet_block(blk, odd, emb) is None
    assert blk(odd, emb).shape == odd.shape


def test_unet_gradients_nodes_on_vs_off():
    """Reduced CFG-DDPM U-Net, eps-MSE loss, gradients written into the flat arena: block nodes on vs off."""
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.DDPM.functions.losses import loss_registry_conditional
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model, ResnetBlock
    from unlearn_saliency_amd.flat import arena_of
    cfg = ddpm_small_config()
    betas = torch.linspace(1e-4, 0.02, 1000, device="cuda")
    xb, cb = ddpm_batch(4, 77)
    x0, c = torch.from_numpy(xb).float().cuda() * 2 - 1, torch.from_numpy(cb).cuda()
    t = torch.tensor([5, 300, 640, 999], device="cuda")
    e = dev_normal(tuple(x0.shape), 78)
    res = []
    for nodes in (True, False):
        model = fill_params(Conditional_Model(cfg), 7000).cuda().train()
        use_salun_convs(model)
        blocks = [m for m in model.modules() if isinstance(m, ResnetBlock)]
        assert blocks and all(b.fused_node for b in blocks)
        for b in blocks:
            b.fused_node = nodes
        arena = arena_of(model)
        arena.grads.zero_()
        for rep in range(2):   # two backward passes accumulate, as the remain + forget passes of one step do
            loss = loss_registry_conditional["simple"](model, x0, t, c, e, betas, cond_drop_prob=0.0)
            loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), arena.grads.clone()))
    (l1, g1), (l2, g2) = res
    assert abs(l1 - l2) <= 1e-6 * abs(l2), (l1, l2)
    err = rel(g1, g2)
    print(f"reduced U-Net: loss {l1:.6f} / {l2:.6f}, flat gradient nodes-on vs nodes-off {err:.2e} of scale")
    assert err <= TOL, err
    assert float(g1.abs().max()) > 0
