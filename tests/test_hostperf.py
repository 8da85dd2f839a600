"""Host-side helpers of the training loops (fastfn.FastFunction, hostperf.freeze_gc, ops._stream_handle): CPU-only checks
that they behave like the stock routes they replace."""
import gc

import torch

from unlearn_saliency_amd import hostperf
from unlearn_saliency_amd.fastfn import FastFunction


class _Scale(FastFunction):
    @staticmethod
    def forward(ctx, a, k, extra=None):
        ctx.k = k
        ctx.save_for_backward(a)
        return a * k if extra is None else a * k + extra

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        return g * ctx.k, None, (g if ctx.needs_input_grad[2] else None)


class _ScaleStock(torch.autograd.Function):
    forward = _Scale.forward
    backward = _Scale.backward


def test_fast_function_matches_the_stock_apply_including_none_and_non_tensor_arguments():
    x = torch.randn(5, 3, dtype=torch.float64, requires_grad=True)
    e = torch.randn(5, 3, dtype=torch.float64, requires_grad=True)
    for args in ((x, 3.0, None), (x, -2.0, None), (x, 0.5, e)):
        outs = []
        for fn in (_Scale, _ScaleStock):
            for t in (x, e):
                t.grad = None
            y = fn.apply(*args)
            y.square().sum().backward()
            outs.append((y.detach().clone(), x.grad.clone(), None if e.grad is None else e.grad.clone()))
        (y1, gx1, ge1), (y0, gx0, ge0) = outs
        assert torch.equal(y1, y0) and torch.equal(gx1, gx0)
        assert (ge1 is None) == (ge0 is None) and (ge1 is None or torch.equal(ge1, ge0))
    # no graph is recorded under no_grad, exactly as with the stock route
    with torch.no_grad():
        assert not _Scale.apply(x, 2.0).requires_grad


def test_freeze_gc_is_idempotent_and_releases_what_an_earlier_call_froze():
    class Node:
        pass
    a, b = Node(), Node()
    a.other, b.other = b, a          # a reference cycle: only the cyclic collector can free it
    hostperf.freeze_gc()
    frozen = gc.get_freeze_count()
    assert frozen > 0
    del a, b                          # garbage now, but frozen: the collector does not look at it
    hostperf.freeze_gc()              # unfreeze -> collect -> freeze: the cycle is gone
    assert gc.get_freeze_count() <= frozen
    gc.unfreeze()


def test_stream_handle_falls_back_without_a_device():
    from unlearn_saliency_amd import ops
    assert callable(ops._stream_handle)   # (the handle itself needs a ROCm device; the -m gpu tests call it on every launch)
