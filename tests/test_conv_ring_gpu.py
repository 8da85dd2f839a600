"""K8r — the LDS-DMA ring convolution (csrc/salun_conv_ring.hip): 3x3 / stride 1 / pad 1 forward and backward-data read
packed weight images.  Checked (a) against PyTorch's fp32 convolution on the CPU (tolerance as tests/test_conv_gpu.py:
1e-5-class, fp32 FMA chains in another summation order), (b) BIT-identical against conv_igemm (same chains, same order —
the claim the kernel's header makes), for every tile variant, ragged batches / channel counts, the epilogue terms and the
in-place backward-data addend; (c) the image cache (ringpack.py): one pack launch per optimizer step for a whole model,
re-packed on raw-pointer updates, torch writes and after a flat arena re-homed the parameters."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, C, H, K)
SHAPES = [
    (64, 64, 32, 64),    # ResNet layer1 (conv_igemm unsplit: the bit-for-bit comparison applies)
    (8, 64, 32, 64),
    (64, 128, 16, 128),
    (16, 256, 8, 256),
    (32, 512, 4, 512),
    (3, 64, 8, 64),      # ragged: N not a multiple of the images per tile
    (4, 384, 16, 256),   # DDPM up-block with concatenated skip
    (2, 40, 16, 72),     # channel counts that are not multiples of 32 (rows past K masked; C % 8 == 0)
    (5, 8, 4, 8),        # one chunk, one partial row tile, ragged N on the 4x4 level
    (2, 128, 32, 3),     # DDPM conv_out (K = 3): forward only (backward-data would reduce over 3 channels)
]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


@pytest.mark.parametrize("N,C,H,K", SHAPES)
@pytest.mark.parametrize("cfg", [1, 2, 3, 5])
def test_ring_matches_torch_cpu_and_conv_igemm_bit_for_bit(N, C, H, K, cfg):
    from unlearn_saliency_amd import ops
    x, w, b = _rand((N, C, H, H), 1), _rand((K, C, 3, 3), 2, 0.1), _rand((K,), 3)
    y_ref = F.conv2d(x, w, b, 1, 1)
    dy = _rand(tuple(y_ref.shape), 4)
    xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    tol = lambda ref: 6e-6 * float(ref.abs().max())
    imf = ops.conv3x3_pack(wd, False)
    assert imf is not None
    y = ops.conv3x3_packed(xd, imf, K, bias=bd, cfg=cfg)
    if y is None:
        pytest.skip("this tile does not divide the pixel space of the shape")
    assert torch.allclose(y.cpu(), y_ref, rtol=3e-6, atol=tol(y_ref))
    # conv_igemm splits the reduction of under-filled launches over workgroups (another summation order): the bit-for-bit
    # claim is about its unsplit form
    unsplit = ops._data_ws_bytes(N, K, H, H, 3, 1) == 0 and ops._data_ws_bytes(N, C, H, H, 3, 1) == 0
    from unlearn_saliency_amd import ringpack
    ringpack.ENABLED[0] = False   # (nothing is registered here; belt and braces: the comparison must be conv_igemm)
    try:
        y_ig = ops.conv2d_forward(xd, wd, bd, 1, 1, H, H)
        dx_ig = ops.conv2d_backward_data(dyd, wd, x.shape, 1, 1) if K % 8 == 0 else None
    finally:
        ringpack.ENABLED[0] = True
    assert torch.equal(y, y_ig) if unsplit else torch.allclose(y, y_ig, rtol=3e-6, atol=tol(y_ref))
    if K % 8 == 0:
        dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, 1, 1)
        imd = ops.conv3x3_pack(wd, True)
        dx = ops.conv3x3_packed(dyd, imd, C, cfg=cfg)
        assert dx is not None and torch.allclose(dx.cpu(), dx_ref, rtol=3e-6, atol=tol(dx_ref))
        assert torch.equal(dx, dx_ig) if unsplit else torch.allclose(dx, dx_ig, rtol=3e-6, atol=tol(dx_ref))
    else:
        assert ops.conv3x3_pack(wd, True) is None


def test_ring_epilogue_terms_and_in_place_addend():
    """bias[k] + nbias[n][k] + addend in the reference's order (DDPM/models/diffusion.py:113-127); backward-data with the
    residual gradient added in place (addend == dx), as resblock.py uses it."""
    from unlearn_saliency_amd import ops
    N, C, H, K = 48, 128, 16, 256
    x, w = _rand((N, C, H, H), 1).cuda(), _rand((K, C, 3, 3), 2, 0.1).cuda()
    b, nb, add = _rand((K,), 3).cuda(), _rand((N, K), 5).cuda(), _rand((N, K, H, H), 6).cuda()
    imf, imd = ops.conv3x3_pack(w, False), ops.conv3x3_pack(w, True)
    y = ops.conv3x3_packed(x, imf, K, bias=b, nbias=nb, addend=add)
    ref = F.conv2d(x.cpu(), w.cpu(), None, 1, 1)
    ref = ((ref + b.cpu()[None, :, None, None]) + nb.cpu()[:, :, None, None]) + add.cpu()
    assert torch.allclose(y.cpu(), ref, rtol=3e-6, atol=6e-6 * float(ref.abs().max()))
    if ops._data_ws_bytes(N, K, H, H, 3, 1) == 0:
        assert torch.equal(y, ops.conv2d_forward(x, w, b, 1, 1, H, H, nbias=nb, addend=add))
    dy = _rand((N, K, H, H), 7).cuda()
    acc = _rand((N, C, H, H), 8).cuda()
    want = ops.conv3x3_packed(dy, imd, C, addend=acc)
    got = ops.conv3x3_packed(dy, imd, C, addend=acc, out=acc)   # in place
    assert got.data_ptr() == acc.data_ptr() and torch.equal(got, want)


def test_under_filled_problems_stay_on_conv_igemm():
    from unlearn_saliency_amd import ops
    x, w = _rand((2, 64, 8, 8), 1).cuda(), _rand((64, 64, 3, 3), 2, 0.1).cuda()
    assert ops.conv3x3_packed(x, ops.conv3x3_pack(w, False), 64) is None          # 2 tiles: the caller's other path
    assert ops.conv3x3_packed(x, ops.conv3x3_pack(w, False), 64, cfg=3) is not None  # pinned: runs anyway


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(16, 32, 3, padding=1, bias=False)
        self.b = nn.Conv2d(32, 32, 3, padding=1, bias=True)
        self.c = nn.Conv2d(32, 8, 1)                       # not a ring layer

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def test_image_cache_one_pack_per_step_and_never_stale():
    from unlearn_saliency_amd import ops, ringpack
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.flat import FlatArena
    torch.manual_seed(0)
    net = _Net().cuda()
    use_salun_convs(net)
    arena = FlatArena.from_module(net)         # re-homes the parameters AFTER registration
    x = torch.randn(64, 16, 32, 32, device="cuda")
    ref = lambda: F.conv2d(torch.relu(F.conv2d(torch.relu(F.conv2d(x, net.a.weight, None, 1, 1)), net.b.weight, net.b.bias, 1, 1)),
                           net.c.weight, net.c.bias)
    n0 = ringpack.PACK_LAUNCHES[0]
    y = net(x)
    assert ringpack.PACK_LAUNCHES[0] == n0 + 1, "both ring layers are packed by ONE launch"
    assert torch.allclose(y, ref(), rtol=1e-4, atol=1e-4)
    y.sum().backward()
    net(x)
    assert ringpack.PACK_LAUNCHES[0] == n0 + 1, "unchanged parameters: no re-pack (forward, backward, forward)"
    # a raw-pointer update (the fused optimizer kernels): PARAM_EPOCH
    g = torch.randn_like(arena.params)
    ops.masked_sgd_step(arena.params, g, None, None, 0.1, 0.0, 0.0, True)
    y2 = net(x)
    assert ringpack.PACK_LAUNCHES[0] == n0 + 2
    assert torch.allclose(y2, ref(), rtol=1e-4, atol=1e-4) and not torch.allclose(y2, y)
    # a torch write on the flat vector, then one on a parameter itself
    with torch.no_grad():
        arena.params.mul_(0.5)
    assert torch.allclose(net(x), ref(), rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        net.a.weight.add_(1.0)
    assert torch.allclose(net(x), ref(), rtol=1e-4, atol=1e-4)
    assert ringpack.PACK_LAUNCHES[0] == n0 + 4
    # unregistered weights (even at a registered weight's old address) are not served
    w_free = torch.randn(32, 16, 3, 3, device="cuda")
    assert ringpack.images(w_free) is None


def test_gradients_through_the_ring_path_match_the_igemm_path():
    from unlearn_saliency_amd import ringpack
    from unlearn_saliency_amd.conv import use_salun_convs
    torch.manual_seed(1)
    net = _Net().cuda()
    use_salun_convs(net)
    x = torch.randn(64, 16, 32, 32, device="cuda", requires_grad=True)
    outs = []
    for on in (True, False):
        ringpack.ENABLED[0] = on
        try:
            net.zero_grad(set_to_none=True)
            x.grad = None
            net(x).square().mean().backward()
            outs.append([x.grad.clone()] + [p.grad.clone() for p in net.parameters()])
        finally:
            ringpack.ENABLED[0] = True
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,C,H,K", [(16, 64, 32, 64), (16, 128, 16, 128), (16, 256, 8, 256), (32, 512, 4, 512),
                                     (3, 64, 8, 64), (4, 384, 16, 256), (2, 320, 32, 320), (5, 64, 4, 96)])
def test_ring_backward_weight_matches_torch_cpu(N, C, H, K):
    """conv3x3_wgrad_ring (the default 3x3 / stride 1 backward-weight when the launch is not flagged as sharing the
    device) against PyTorch on the CPU; deterministic; and the flagged launch (conv_wgrad_v) agrees to rounding."""
    from unlearn_saliency_amd import ops
    x, dy = _rand((N, C, H, H), 1), _rand((N, K, H, H), 4)
    dw_ref = torch.nn.grad.conv2d_weight(x, (K, C, 3, 3), dy, 1, 1)
    xd, dyd = x.cuda(), dy.cuda()
    dw = ops.conv2d_backward_weight(xd, dyd, (K, C, 3, 3), 1, 1)
    tol = 6e-6 * float(dw_ref.abs().max())
    assert dw is not None and torch.allclose(dw.cpu(), dw_ref, rtol=3e-6, atol=tol)
    assert torch.equal(dw, ops.conv2d_backward_weight(xd, dyd, (K, C, 3, 3), 1, 1))
    dws = ops.conv2d_backward_weight(xd, dyd, (K, C, 3, 3), 1, 1, shared=True)
    assert torch.allclose(dws.cpu(), dw_ref, rtol=3e-6, atol=tol)
    acc = _rand((K, C, 3, 3), 9).cuda()
    got = ops.conv2d_backward_weight(xd, dyd, (K, C, 3, 3), 1, 1, out=acc.clone(), accumulate=True)
    assert torch.allclose(got, acc + dw, rtol=1e-6, atol=tol)
