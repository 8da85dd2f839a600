"""Fused BatchNorm2d (+ residual add) (+ ReLU) on the kernels of csrc/salun_norm.hip.

`fused_bn_act(x, bn, residual=None, relu=True)` runs an `nn.BatchNorm2d` module's math (its parameters, running
statistics, momentum, eps, train/eval mode) plus the optional residual add and ReLU that follow it in a
ResNet block as ONE autograd node: 2 forward + 2 backward streaming launches (+ two tiny finalisers) instead of
the library's BN kernels, a ReLU, an add, a ReLU-backward and a gradient add — about half the HBM passes.
`use_fused_bn(model)` switches the CIFAR ResNet blocks of this package to it; state_dict is unchanged.
"""
from __future__ import annotations

import torch

from .fastfn import FastFunction
import torch.nn as nn
import torch.nn.functional as F

from . import gradsink, ops


class _FusedBN(FastFunction):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu, nbt=None):
        out = ops.bn_forward(x, residual, weight, bias, running_mean, running_var, training, momentum, eps, relu, nbt)
        if out is None:
            raise RuntimeError("fused BN: unsupported shape (H*W must be a multiple of 4)")
        y, mean, invstd = out
        ctx.save_for_backward(x, y if relu else None, weight, bias, mean, invstd)
        ctx.params = (weight, bias)  # Parameter objects for gradsink (saved_tensors are detached aliases under checkpointing)
        ctx.cfg = (bool(training), bool(relu), residual is not None)
        ctx.mark_non_differentiable(mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, bias, mean, invstd = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        gw, gb = gradsink.sink(ctx.params[0]), gradsink.sink(ctx.params[1])
        if gw is None or gb is None:
            gw = gb = None
        dx, dres, dgamma, dbeta = ops.bn_backward(dy.contiguous(), y, x, weight, mean, invstd, training, relu,
                                                  has_res and ctx.needs_input_grad[3], gw, gb)
        if gw is not None:
            gradsink.arrived(weight)
            gradsink.arrived(bias)
            dgamma = dbeta = None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None


def _eligible(x: torch.Tensor, bn: nn.BatchNorm2d) -> bool:
    return (type(bn) is nn.BatchNorm2d and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and (x.shape[2] * x.shape[3]) % 4 == 0
            and bn.affine and bn.track_running_stats and bn.momentum is not None and not torch.is_autocast_enabled())


def fused_bn_act(x: torch.Tensor, bn: nn.BatchNorm2d, residual: torch.Tensor = None, relu: bool = True):
    if not _eligible(x, bn):
        y = bn(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    x = x.contiguous()
    if residual is not None:
        residual = residual.contiguous()
    return _FusedBN.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.training,
                          bn.momentum, bn.eps, relu, bn.num_batches_tracked)  # the kernel bumps the counter


class _FusedGN(FastFunction):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu):
        out = ops.gn_forward(x, weight, bias, groups, eps, silu)
        if out is None:
            raise RuntimeError("fused GroupNorm: unsupported shape")
        z, mean, rstd = out
        ctx.save_for_backward(x, weight, bias, mean, rstd)  # y and sigmoid(y) are recomputed in backward
        ctx.params = (weight, bias)
        ctx.cfg = (int(groups), bool(silu))
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        groups, silu = ctx.cfg
        gw, gb = gradsink.sink(ctx.params[0]), gradsink.sink(ctx.params[1])
        if gw is None or gb is None:
            gw = gb = None
        dx, dgamma, dbeta = ops.gn_backward(dz.contiguous(), x, weight, bias, mean, rstd, groups, silu, gw, gb)
        if gw is not None:
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, None


class _FusedGN16(FastFunction):
    """GroupNorm (+ SiLU) of a bf16 activation on csrc/salun_norm_bf16.hip (K12): NHWC bf16 in / out, fp32 statistics.
    The input may be any 4-D bf16 device tensor (logical NCHW); `channels_last` inputs are used as they are."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu):
        xn = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        y, mr, ab = ops.gn_bf16_forward(xn, weight, bias, groups, eps, silu)
        # the parameters ride on the node as plain references, not as saved tensors: under activation checkpointing every
        # saved tensor costs a pack hook in the forward and another in the recompute (~16 us of host time each; the SD
        # step made 2,136 such calls, a fifth of its host time, 40 % of them for parameters) — nothing writes a
        # parameter between a forward and its backward in the loops of this package
        ctx.save_for_backward(xn, mr, ab)
        ctx.params = (weight, bias)
        ctx.cfg = (int(groups), bool(silu))
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xn, mr, ab = ctx.saved_tensors
        weight, bias = ctx.params
        groups, silu = ctx.cfg
        dyn = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        gw, gb = gradsink.sink(ctx.params[0]), gradsink.sink(ctx.params[1])
        sunk = gw is not None and gb is not None
        if not sunk:
            gw, gb = torch.empty_like(weight), torch.empty_like(bias)
        dx = ops.gn_bf16_backward(dyn, xn, weight, mr, ab, groups, silu, gw, gb, accumulate=sunk)
        return dx.permute(0, 3, 1, 2), (None if sunk else gw), (None if sunk else gb), None, None, None


_FUSED_GN = True


def enable_fused_gn(flag: bool) -> None:
    """Process-wide switch of `fused_gn_act` (on by default; off = the library's group_norm + sigmoid + mul)."""
    global _FUSED_GN
    _FUSED_GN = bool(flag)


def fused_gn_act(x: torch.Tensor, gn: nn.GroupNorm, silu: bool = True) -> torch.Tensor:
    """`[x * sigmoid(x)](gn(x))` — one forward and two backward launches on csrc/salun_norm.hip for fp32 NCHW device
    tensors whose H*W is a power of two; anything else runs the library ops."""
    hw = x.shape[2] * x.shape[3] if x.dim() == 4 else 0
    if (_FUSED_GN and type(gn) in _GN_TYPES and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and gn.affine
            and x.shape[1] % 8 == 0 and gn.weight.dtype == torch.float32):
        return _FusedGN16.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, silu)
    if (_FUSED_GN and type(gn) in _GN_TYPES and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and gn.affine and hw >= 4 and (hw & (hw - 1)) == 0 and x.shape[1] // gn.num_groups <= 256
            and not torch.is_autocast_enabled()):
        return _FusedGN.apply(x.contiguous(), gn.weight, gn.bias, gn.num_groups, gn.eps, silu)
    y = gn(x)
    return y * torch.sigmoid(y) if silu else y


_GN_TYPES = {nn.GroupNorm}


def use_fused_bn(model: nn.Module, blocks: bool = True) -> int:
    """Switch this package's CIFAR ResNet (stem + BasicBlocks) to the fused BN path; with `blocks` each BasicBlock
    additionally runs as one autograd node (resblock.py).  Returns the number of BatchNorm layers covered."""
    from .Classification.models import resnet_cifar as R
    n = 0
    for mod in model.modules():
        if isinstance(mod, R.BasicBlock):
            mod.fused_bn = True
            mod.fused_block = bool(blocks)
            n += 2 + (1 if mod.downsample is not None else 0)
        elif isinstance(mod, R.ResNetCifar):
            mod.fused_bn = True
            n += 1
    return n
