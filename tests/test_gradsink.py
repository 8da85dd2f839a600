"""Engine behaviour that gradsink.py relies on (CPU): a leaf whose custom Function returns None for its gradient
still gets its post-accumulate-grad hooks fired exactly once per backward, and sink() declines whenever writing
into .grad directly would not be equivalent to autograd's own accumulation."""
import torch

from unlearn_saliency_amd import gradsink


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x * w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        if w.grad is not None:  # what the HIP kernels do: add into the existing storage, hand autograd nothing
            w.grad.add_(g * x)
            return g * w, None
        return g * w, g * x


def test_post_accumulate_hooks_fire_once_even_when_function_returns_none():
    w = torch.nn.Parameter(torch.full((3,), 2.0))
    x = torch.ones(3, requires_grad=True)
    fired = []
    w.register_post_accumulate_grad_hook(lambda p: fired.append(p.grad.clone()))
    _Mul.apply(x, w).sum().backward()  # first: autograd route (no .grad yet)
    _Mul.apply(x, w).sum().backward()  # second: direct accumulation, None returned
    assert len(fired) == 2
    assert torch.equal(fired[0], torch.ones(3)) and torch.equal(fired[1], torch.full((3,), 2.0))


def test_sink_declines_when_not_equivalent():
    p = torch.nn.Parameter(torch.zeros(4))
    with torch.no_grad():
        assert gradsink.sink(p) is None  # no .grad
        p.grad = torch.zeros(4)
        assert gradsink.sink(p) is None  # host tensor: the kernels only write device memory
        assert gradsink.sink(None) is None and gradsink.sink(torch.zeros(4)) is None
    assert gradsink.sink(p) is None  # grad mode on (double backward)
