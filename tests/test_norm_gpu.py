"""Fused BatchNorm(+add)(+ReLU) kernels vs PyTorch's fp32 reference ops (training and eval mode), and the
ResNet-18 module switch end to end."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(256, 64, 32, 32), (32, 128, 16, 16), (16, 256, 8, 8), (64, 512, 4, 4), (3, 8, 2, 2)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_fused_bn_matches_torch(shape, training, relu, res):
    from unlearn_saliency_amd.norm import fused_bn_act
    torch.manual_seed(0)
    N, C, H, W = shape
    x = (torch.randn(shape, device="cuda") * 2 + 0.5).requires_grad_(True)
    r = torch.randn(shape, device="cuda").requires_grad_(True) if res else None
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    ref = nn.BatchNorm2d(C).cuda()
    ref.load_state_dict(bn.state_dict())
    bn.train(training); ref.train(training)
    y = fused_bn_act(x, bn, residual=r, relu=relu)
    x2 = x.detach().clone().requires_grad_(True)
    r2 = r.detach().clone().requires_grad_(True) if res else None
    y2 = ref(x2)
    if res:
        y2 = y2 + r2
    z2 = y2
    if relu:
        y2 = F.relu(y2)
    assert torch.allclose(y, y2, rtol=3e-5, atol=3e-6)  # every tolerance here: 3x the worst use measured on the MI355X
    # pre-activations within rounding of 0 may sit on different sides of the ReLU in the two computations:
    # give those (a handful out of millions) no upstream gradient so the comparison is well defined
    dy = torch.randn_like(y) * (z2.detach().abs() > 1e-5)
    y.backward(dy); y2.backward(dy)
    tol = lambda t: 1e-6 * float(t.abs().max()) + 5e-8
    assert torch.allclose(x.grad, x2.grad, rtol=5e-6, atol=tol(x2.grad))
    assert torch.allclose(bn.weight.grad, ref.weight.grad, rtol=5e-6, atol=tol(ref.weight.grad))
    assert torch.allclose(bn.bias.grad, ref.bias.grad, rtol=5e-6, atol=tol(ref.bias.grad))
    if res:
        assert torch.allclose(r.grad, r2.grad, rtol=1e-5, atol=1e-6)
    if training:
        assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-6, atol=1e-8)
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("blocks", [False, True])
@pytest.mark.parametrize("train", [True, False])
def test_resnet18_fused_bn_matches_unfused(train, blocks):
    """Whole model: the fused path is as close to a float64 evaluation as the unfused fp32 path is (fp32 rounding
    through 20 train-mode BN layers is ~1e-2 relative either way, so the two fp32 paths are not compared directly)."""
    from unlearn_saliency_amd.Classification.models import model_dict
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.norm import use_fused_bn
    torch.manual_seed(0)
    a = model_dict["resnet18"](num_classes=10).cuda()
    b = model_dict["resnet18"](num_classes=10).cuda()
    d = model_dict["resnet18"](num_classes=10).cuda().double()
    b.load_state_dict(a.state_dict())
    d.load_state_dict(a.state_dict())
    use_salun_convs(a); use_salun_convs(b)
    assert use_fused_bn(b, blocks=blocks) == 20
    x = torch.rand(64, 3, 32, 32, device="cuda")
    y = torch.randint(0, 10, (64,), device="cuda")
    losses = []
    for m, xx in ((a, x), (b, x), (d, x.double())):
        m.train(train)
        loss = F.cross_entropy(m(xx), y)
        loss.backward()
        losses.append(loss.item())
    assert abs(losses[1] - losses[2]) <= 1e-5 * abs(losses[2])
    worst_a = worst_b = 0.0
    for (k, p), q, r in zip(a.named_parameters(), b.parameters(), d.parameters()):
        den = float(r.grad.abs().max()) + 1e-12
        worst_a = max(worst_a, float((p.grad.double() - r.grad).abs().max()) / den)
        worst_b = max(worst_b, float((q.grad.double() - r.grad).abs().max()) / den)
    assert worst_b < 3 * worst_a + 1e-5, (worst_a, worst_b)
    for (k, u), v in zip(d.named_buffers(), b.buffers()):
        assert torch.allclose(u.float(), v.float(), rtol=1e-6, atol=1e-7), k
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())


@pytest.mark.parametrize("shape", [(8, 64, 64, 3, 1, 1, 32), (8, 64, 128, 3, 2, 1, 32), (8, 64, 128, 1, 2, 0, 32),
                                   (4, 256, 512, 3, 2, 1, 8), (4, 512, 512, 3, 1, 1, 4)])
def test_backward_data_addend_is_a_fused_add(shape):
    """dx = dgrad(dy, w) + addend, bit-identical to the separate add (one fp32 add per element either way)."""
    from unlearn_saliency_amd import ops
    N, C, K, R, s, pad, H = shape
    torch.manual_seed(1)
    P = (H + 2 * pad - R) // s + 1
    dy = torch.randn(N, K, P, P, device="cuda")
    w = torch.randn(K, C, R, R, device="cuda") * 0.05
    add = torch.randn(N, C, H, H, device="cuda")
    base = ops.conv2d_backward_data(dy, w, (N, C, H, H), s, pad)
    fused = ops.conv2d_backward_data(dy, w, (N, C, H, H), s, pad, addend=add)
    assert base is not None and fused is not None
    assert torch.equal(fused, base + add)


def test_direct_grad_accumulation_equals_autograd_route():
    """Kernels adding into .grad (gradsink) == returning gradients for AccumulateGrad, over two accumulating
    backward passes; post-accumulate hooks fire once per parameter per backward either way."""
    from unlearn_saliency_amd import gradsink
    from unlearn_saliency_amd.Classification.models import model_dict
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.flat import FlatArena
    from unlearn_saliency_amd.norm import use_fused_bn
    torch.manual_seed(0)
    x = torch.rand(32, 3, 32, 32, device="cuda")
    y = torch.randint(0, 10, (32,), device="cuda")
    results = []
    for direct in (True, False):
        torch.manual_seed(0)
        m = model_dict["resnet18"](num_classes=10).cuda()
        use_salun_convs(m); use_fused_bn(m)
        m.eval()  # deterministic forward across the two passes
        arena = FlatArena.from_module(m)
        fired = {}
        for n, p in m.named_parameters():
            p.register_post_accumulate_grad_hook(lambda _p, n=n: fired.__setitem__(n, fired.get(n, 0) + 1))
        gradsink.enable(direct)
        try:
            arena.zero_grad()
            for _ in range(2):
                F.cross_entropy(m(x), y).backward()
        finally:
            gradsink.enable(True)
        assert set(fired) == {n for n, _ in m.named_parameters()}
        assert set(fired.values()) == {2}, (direct, {n: c for n, c in fired.items() if c != 2})
        results.append(arena.grads.clone())
    a, b = results
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))


@pytest.mark.parametrize("shape,groups", [((8, 128, 32, 32), 32), ((4, 256, 16, 16), 32), ((4, 384, 16, 16), 32),
                                          ((6, 512, 4, 4), 32), ((3, 64, 2, 2), 32), ((2, 320, 64, 64), 32),
                                          ((2, 1920, 32, 32), 32), ((5, 96, 8, 8), 8)])
@pytest.mark.parametrize("silu", [True, False])
def test_fused_group_norm_matches_torch(shape, groups, silu):
    """GroupNorm(+SiLU) kernels vs torch's fp32 ops, forward and backward (cached and re-read modes, all segment
    widths: H*W/4 from 1 to 1024 lanes per channel)."""
    from unlearn_saliency_amd.norm import fused_gn_act
    torch.manual_seed(0)
    N, C, H, W = shape
    x = (torch.randn(shape, device="cuda") * 1.5 + 0.3).requires_grad_(True)
    gn = nn.GroupNorm(groups, C, eps=1e-6).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5); gn.bias.normal_()
    ref = nn.GroupNorm(groups, C, eps=1e-6).cuda()
    ref.load_state_dict(gn.state_dict())
    z = fused_gn_act(x, gn, silu=silu)
    x2 = x.detach().clone().requires_grad_(True)
    y2 = ref(x2)
    z2 = y2 * torch.sigmoid(y2) if silu else y2
    assert torch.allclose(z, z2, rtol=1e-5, atol=2e-6)
    dz = torch.randn_like(z)
    z.backward(dz); z2.backward(dz)
    tol = lambda t: 5e-7 * float(t.abs().max()) + 2.5e-8
    assert torch.allclose(x.grad, x2.grad, rtol=2.5e-6, atol=tol(x2.grad))
    assert torch.allclose(gn.weight.grad, ref.weight.grad, rtol=2.5e-6, atol=tol(ref.weight.grad))
    assert torch.allclose(gn.bias.grad, ref.bias.grad, rtol=2.5e-6, atol=tol(ref.bias.grad))
    # second backward accumulates into .grad in the kernel (gradsink) — equals doubling
    g1 = gn.weight.grad.clone()
    fused_gn_act(x.detach().requires_grad_(True), gn, silu=silu).backward(dz)
    assert torch.allclose(gn.weight.grad, 2 * g1, rtol=1e-5, atol=tol(g1))
    # deterministic
    a = fused_gn_act(x.detach(), gn, silu=silu)
    assert torch.equal(a, fused_gn_act(x.detach(), gn, silu=silu))
