"""MFMA convolution kernels (forward / backward-data / backward-weight) vs PyTorch's fp32 convolution
on the CPU, for every layer shape of ResNet-18 (CIFAR) and the DDPM U-Net families, plus the module swap.
fp32 FMA chains in a different summation order: tolerance 1e-5 of the tensor's scale (north_star: 1e-5 rel)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, C, H, K, R, stride, pad)
SHAPES = [
    (8, 3, 32, 64, 3, 1, 1),      # ResNet stem (C = 3: ragged reduction chunk)
    (8, 64, 32, 64, 3, 1, 1),     # layer1
    (8, 64, 32, 128, 3, 2, 1),    # layer2.0.conv1
    (8, 64, 32, 128, 1, 2, 0),    # layer2.0.downsample
    (8, 128, 16, 128, 3, 1, 1),
    (16, 128, 16, 256, 3, 2, 1),
    (16, 128, 16, 256, 1, 2, 0),
    (16, 256, 8, 256, 3, 1, 1),
    (32, 256, 8, 512, 3, 2, 1),
    (32, 256, 8, 512, 1, 2, 0),
    (32, 512, 4, 512, 3, 1, 1),
    (3, 64, 8, 64, 3, 1, 1),      # ragged: N not a multiple of the images-per-tile
    (4, 256, 16, 128, 1, 1, 0),   # DDPM nin_shortcut / attention projections
    (4, 384, 16, 256, 3, 1, 1),   # DDPM up-block with concatenated skip
    (4, 128, 32, 3, 3, 1, 1),     # DDPM conv_out (K = 3: masked channel tile)
    (2, 40, 16, 72, 3, 1, 1),     # channel counts that are not multiples of 32
    (8, 3, 32, 64, 3, 2, 1),      # small-C backward-weight kernel, stride 2
    (5, 1, 16, 40, 3, 1, 1),      # single input channel, ragged N and K
    (6, 4, 64, 320, 3, 1, 1),     # SD conv_in (C = 4 latents)
    (2, 320, 32, 320, 3, 1, 1),   # SD: K = 320 is not a multiple of the 128-channel tile (ragged block, fast staging)
    (2, 64, 16, 200, 1, 1, 0),    # ragged K, 1x1
]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


@pytest.mark.parametrize("N,C,H,K,R,stride,pad", SHAPES)
def test_conv_kernels_match_torch_cpu(N, C, H, K, R, stride, pad):
    from unlearn_saliency_amd import ops
    x = _rand((N, C, H, H), 1)
    w = _rand((K, C, R, R), 2, 0.1)
    b = _rand((K,), 3)
    P = (H + 2 * pad - R) // stride + 1
    y_ref = F.conv2d(x, w, b, stride, pad)
    dy = _rand(tuple(y_ref.shape), 4)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, stride, pad)
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride, pad)
    xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    y = ops.conv2d_forward(xd, wd, bd, stride, pad, P, P)
    assert y is not None, "shape unexpectedly outside the tiling domain"
    tol = lambda ref: 6e-6 * float(ref.abs().max())  # rtol / atol: 3x the worst use measured (tools/measure_tolerances.py)
    assert torch.allclose(y.cpu(), y_ref, rtol=3e-6, atol=tol(y_ref))
    dx = ops.conv2d_backward_data(dyd, wd, x.shape, stride, pad)
    assert dx is not None and torch.allclose(dx.cpu(), dx_ref, rtol=3e-6, atol=tol(dx_ref))
    dw = ops.conv2d_backward_weight(xd, dyd, w.shape, stride, pad)
    assert dw is not None and torch.allclose(dw.cpu(), dw_ref, rtol=3e-6, atol=tol(dw_ref))
    # deterministic run to run
    assert torch.equal(dw, ops.conv2d_backward_weight(xd, dyd, w.shape, stride, pad))
    assert torch.equal(y, ops.conv2d_forward(xd, wd, bd, stride, pad, P, P))


def test_asymmetric_padding_downsample():
    """DDPM Downsample: pad (0,1,0,1) then 3x3 stride 2 (DDPM/models/diffusion.py:75-79)."""
    from unlearn_saliency_amd.conv import conv2d_lowpad
    x = _rand((4, 128, 32, 32), 5).cuda().requires_grad_(True)
    w = _rand((128, 128, 3, 3), 6, 0.05).cuda().requires_grad_(True)
    b = _rand((128,), 7).cuda().requires_grad_(True)
    y = conv2d_lowpad(x, w, b, 2, 0, 16, 16)
    y.square().sum().backward()
    x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    y2 = F.conv2d(F.pad(x2, (0, 1, 0, 1)), w2, b2, stride=2, padding=0)
    y2.square().sum().backward()
    for a, r in ((y, y2), (x.grad, x2.grad), (w.grad, w2.grad), (b.grad, b2.grad)):
        assert torch.allclose(a, r, rtol=5e-6, atol=1.5e-6 * float(r.abs().max()))


def test_unsupported_shape_falls_back():
    from unlearn_saliency_amd import ops
    x = torch.randn(2, 8, 12, 12, device="cuda")  # width 12 is not a power of two
    w = torch.randn(8, 8, 3, 3, device="cuda")
    assert ops.conv2d_forward(x, w, None, 1, 1, 12, 12) is None


def test_module_swap_resnet18_forward_backward():
    """Whole-network check against a float64 CPU reference (direct convolution): the MFMA path must be within
    fp32 round-off of it.  (The library path may use Winograd transforms, so it is not the yardstick.)"""
    from unlearn_saliency_amd.Classification.models import model_dict
    from unlearn_saliency_amd.conv import SalunConv2d, use_salun_convs
    torch.manual_seed(0)
    ref = model_dict["resnet18"](num_classes=10)
    mine = model_dict["resnet18"](num_classes=10)
    mine.load_state_dict(ref.state_dict())
    mine.cuda()
    n = use_salun_convs(mine)
    assert n == 20 and isinstance(mine.conv1, SalunConv2d)
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    x = torch.rand(16, 3, 32, 32)
    y = torch.randint(0, 10, (16,))
    ref.double().train()
    F.cross_entropy(ref(x.double()), y).backward()
    mine.train()
    F.cross_entropy(mine(x.cuda()), y.cuda()).backward()
    # the library fp32 path on the same inputs, as a yardstick for what fp32 round-off does through 20 BN layers
    lib = model_dict["resnet18"](num_classes=10)
    lib.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    lib.cuda().train()
    F.cross_entropy(lib(x.cuda()), y.cuda()).backward()
    worst = worst_lib = 0.0
    for (k, p), q, l in zip(ref.named_parameters(), mine.parameters(), lib.parameters()):
        scale = float(p.grad.abs().max())
        err = float((p.grad.float() - q.grad.cpu()).abs().max()) / scale
        err_lib = float((p.grad.float() - l.grad.cpu()).abs().max()) / scale
        worst, worst_lib = max(worst, err), max(worst_lib, err_lib)
    print(f"worst gradient error relative to tensor scale vs float64: mfma {worst:.3e}, library {worst_lib:.3e}")
    # fp32 round-off through 20 train-mode BN layers dominates both; the MFMA path must be no worse than the library's
    assert worst < 3 * worst_lib + 1e-5


@pytest.mark.parametrize("N,K,H", [(128, 128, 32), (5, 3, 16), (8, 200, 4), (2, 40, 7), (300, 64, 8)])
def test_channel_sum_is_the_bias_gradient(N, K, H):
    """`salun_channel_sum` (the conv-bias gradient, round 3) vs a float64 sum on the host; accumulate adds in place;
    deterministic.  Tolerance 2e-6 of the sum of |dy| per channel (fp32 partial sums in a fixed tree)."""
    from unlearn_saliency_amd import ops
    dy = _rand((N, K, H, H), 9)
    ref = dy.double().sum(dim=(0, 2, 3))
    scale = dy.double().abs().sum(dim=(0, 2, 3))
    d = dy.cuda()
    got = ops.channel_sum(d)
    assert got.shape == (K,) and torch.all((got.cpu().double() - ref).abs() <= 2e-6 * scale)
    assert torch.equal(got, ops.channel_sum(d))
    base = _rand((K,), 10).cuda()
    acc = base.clone()
    ops.channel_sum(d, out=acc, accumulate=True)
    assert torch.equal(acc, base + got)
    view = d[:, :, :, :].permute(0, 1, 2, 3)[:, :, 1:, :]  # odd offset: 16-byte alignment lost -> scalar path
    if H > 1:
        v = view.contiguous()
        assert torch.allclose(ops.channel_sum(v).cpu().double(), v.cpu().double().sum(dim=(0, 2, 3)), rtol=6e-6, atol=6e-5)


def test_conv_bias_gradient_lands_in_the_flat_arena_without_autograd_adds():
    """A biased SalunConv2d inside a FlatArena: after backward, bias.grad (a view of the flat gradient) holds the
    channel sums of dy, accumulated over two backward passes, and equals the library module's bias gradient."""
    import copy
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.flat import arena_of
    torch.manual_seed(4)
    lib = torch.nn.Sequential(torch.nn.Conv2d(32, 64, 3, padding=1), torch.nn.SiLU(), torch.nn.Conv2d(64, 32, 1)).cuda()
    own = copy.deepcopy(lib)
    assert use_salun_convs(own) == 2
    arena = arena_of(own)
    x = torch.randn(8, 32, 16, 16, device="cuda")
    arena.zero_grad()
    for _ in range(2):
        own(x).square().mean().backward()
        lib(x).square().mean().backward()
    assert own[0].bias.grad.data_ptr() == arena.grads[arena.offsets[1]:].data_ptr()
    for a, b in zip(own.parameters(), lib.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=6.6e-5, atol=6.6e-7 * float(b.grad.abs().max())), a.shape
