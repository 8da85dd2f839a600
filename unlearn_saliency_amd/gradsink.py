"""Writing parameter gradients straight into their `.grad` storage from inside a custom backward.

With the flat arena (flat.py) every parameter's `.grad` is a persistent view of one fp32 vector that is zeroed
once per step.  autograd's own route — return dW from backward, AccumulateGrad does `.grad += dW` — costs one extra
elementwise launch per parameter per step (62 for ResNet-18, 334 for the DDPM U-Net).  The kernels that produce
weight gradients here can accumulate into a destination themselves (`salun_conv2d_backward_weight(accumulate=1)`,
`salun_bn_backward(grad_*_acc)`), so the custom Functions ask `sink(p)` for the destination, let the kernel add
into it and return None to autograd; post-accumulate-grad hooks (the bucketed all-reduce of
dist.BucketedGradReducer) still fire exactly once per parameter (see `arrived`).

`sink(p)` is None — and the Functions fall back to returning the gradient — whenever that shortcut is not
provably equivalent: no `.grad` yet, a non-contiguous / non-fp32 `.grad`, gradient hooks on the tensor, or a
double-backward (`torch.is_grad_enabled()` inside backward means create_graph=True).
"""
from __future__ import annotations

import torch

_enabled = True


def enable(flag: bool) -> None:
    global _enabled
    _enabled = bool(flag)


def sink(p):
    if not _enabled or p is None or not isinstance(p, torch.nn.Parameter) or torch.is_grad_enabled():
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or not g.is_cuda:
        return None
    if p._backward_hooks:  # tensor hooks would see / rewrite the incoming gradient: keep autograd's route
        return None
    return g


def arrived(p) -> None:
    """Nothing to do: autograd still runs the leaf's AccumulateGrad node when the Function returns None for it, and
    that node fires the post-accumulate-grad hooks (after our kernels were enqueued, so stream order holds).
    tests/test_gradsink.py pins this engine behaviour; if it ever changes, fire the hooks here."""
    return None
