"""`nn.Conv2d` on the hand-written MFMA convolution kernels (csrc/salun_conv.hip).

`use_salun_convs(model)` re-classes every eligible `nn.Conv2d` of a model to `SalunConv2d` in place —
parameters, names and state_dict are untouched (the weights stay views of the flat arena) — so forward,
backward-data and backward-weight run as `salun_conv2d_*` launches instead of whatever the library's
heuristics pick for fp32 on gfx950 (`naive_conv_*` at > 1 s per ResNet-18 step on a cold
find-db).  Shapes outside the kernels' tiling domain, non-fp32 inputs and autocast regions fall back to the
library convolution per call — LOUDLY: every such call is counted in `LIBRARY_CONV_CALLS` (reported by the
benchmarks as `library_conv_calls`), and `strict(True)` turns the fallback into an error.
"""
from __future__ import annotations

import os

import torch

from .fastfn import FastFunction
import torch.nn as nn
import torch.nn.functional as F

from . import dist as sdist
from . import gradsink, ops, resblock, ringpack


# calls that went to the library (MIOpen) instead of the MFMA kernels, by reason
LIBRARY_CONV_CALLS = {"shape": 0, "dtype_or_autocast": 0, "backward": 0}
_STRICT = [False]


# SALUN_BLOCK_NODES=0: keep the diffusion ResnetBlocks as separate autograd nodes (A/B switch for the benchmarks)
_BLOCK_NODES = [os.environ.get("SALUN_BLOCK_NODES", "1") != "0"]
# SALUN_OWN_GEMM=0: keep the diffusion U-Nets' Linear layers and fp32 attention on the library (A/B switch)
_OWN_GEMM = [os.environ.get("SALUN_OWN_GEMM", "1") != "0"]


# SALUN_RING=0: keep every 3x3 convolution on conv_igemm (A/B switch for the benchmarks)
_RING = [os.environ.get("SALUN_RING", "1") != "0"]


def strict(on: bool = True) -> None:
    """strict(True): a convolution that would fall back to the library raises instead."""
    _STRICT[0] = bool(on)


def library_conv_calls() -> int:
    return sum(LIBRARY_CONV_CALLS.values())


def reset_library_conv_calls() -> None:
    for k in LIBRARY_CONV_CALLS:
        LIBRARY_CONV_CALLS[k] = 0


def _fallback(reason: str, what: str) -> None:
    LIBRARY_CONV_CALLS[reason] += 1
    if _STRICT[0]:
        raise RuntimeError(f"salun conv: {what} would run on the library convolution ({reason}) and strict mode is on")


def _eligible(mod: nn.Conv2d) -> bool:
    k, s, p, d = mod.kernel_size, mod.stride, mod.padding, mod.dilation
    return (isinstance(p, tuple) and k[0] == k[1] and k[0] in (1, 3) and s[0] == s[1] and s[0] in (1, 2)
            and p[0] == p[1] and p[0] <= k[0] - 1 and d == (1, 1) and mod.groups == 1
            and mod.padding_mode == "zeros" and mod.weight.dtype == torch.float32)


class _ConvFn(FastFunction):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, P, Q):
        y = ops.conv2d_forward(x, w, bias, stride, pad, P, Q)
        if y is None:  # outside the tiling domain
            _fallback("shape", f"forward {tuple(x.shape)} * {tuple(w.shape)} stride {stride} pad {pad}")
            y = F.conv2d(x, w, bias, stride, pad)
            ctx.native = False
        else:
            ctx.native = True
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.pad, ctx.has_bias = stride, pad, bias is not None
        # the parameter OBJECTS, only looked at for their .grad destinations (gradsink), never read as values: under
        # activation checkpointing `saved_tensors` hands back detached aliases, which are not the nn.Parameters any more
        ctx.bias_param, ctx.weight_param = bias, w
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        stride, pad = ctx.stride, ctx.pad
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dst = gradsink.sink(ctx.weight_param) if ctx.native else None
            # the weight gradient goes straight into .grad (gradsink): nothing on the main stream consumes it before
            # the end of the backward pass, so it can run on the side stream next to backward-data (resblock.py)
            # (also under data parallel: the slice's all-reduce waits for the side stream, dist.BucketedGradReducer)
            overlap = (dst is not None and resblock.OVERLAP_WGRAD)
            if overlap:
                main, side = torch.cuda.current_stream(dy.device), resblock._side_stream(dy.device)
                side.wait_stream(main)
                with resblock._on_side(side, dst is None):
                    dw = ops.conv2d_backward_weight(x, dy, w.shape, stride, pad, out=dst, accumulate=True, shared=True)
                if dw is None:  # outside the kernel's domain after all: nothing was launched
                    overlap = False
                else:
                    x.record_stream(side)
                    dy.record_stream(side)
                    resblock.hold_until_join(dy)  # autograd must not accumulate into dy in place while the side stream reads it
                    resblock._join_at_end_of_backward(dy.device)
            if not overlap:
                dw = ops.conv2d_backward_weight(x, dy, w.shape, stride, pad, out=dst, accumulate=True) if ctx.native else None
            if dw is None:
                _fallback("backward", f"backward-weight {tuple(x.shape)} * {tuple(w.shape)}")
                dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride, pad)
            elif dst is not None:  # already added into w.grad by the kernel
                gradsink.arrived(w)
                dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_backward_data(dy, w, x.shape, stride, pad) if ctx.native else None
            if dx is None:
                _fallback("backward", f"backward-data {tuple(x.shape)} * {tuple(w.shape)}")
                dx = torch.nn.grad.conv2d_input(x.shape, w, dy, stride, pad)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # own streaming kernel instead of ATen's generic reduction; added straight into bias.grad when that is a
            # plain view of the flat arena (no AccumulateGrad launch)
            bdst = gradsink.sink(ctx.bias_param)
            if bdst is not None:
                ops.channel_sum(dy, out=bdst, accumulate=True)
                gradsink.arrived(ctx.bias_param)
            else:
                db = ops.channel_sum(dy)
        return dx, dw, db, None, None, None, None


class SalunConv2d(nn.Conv2d):
    """Same parameters / state_dict as nn.Conv2d; fp32 NCHW device tensors go through the MFMA kernels."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_autocast_enabled():
            # NCHW-contiguous is the kernels' layout; a strided view (e.g. the channels-last result of an attention
            # reshape) is materialised once here rather than sending the whole rest of the network down the library path
            x = x.contiguous()
            R, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
            P = (x.shape[2] + 2 * p - R) // s + 1
            Q = (x.shape[3] + 2 * p - R) // s + 1
            return _ConvFn.apply(x, self.weight, self.bias, s, p, P, Q)
        _fallback("dtype_or_autocast", f"forward of a {x.dtype} / autocast input")
        return super().forward(x)


def conv2d_lowpad(x, weight, bias, stride, pad_lo, P, Q):
    """Functional form with explicit low-side padding and output size (asymmetric padding such as the DDPM
    downsampler's (0,1,0,1): pad_lo = 0 with P, Q computed for the padded extent)."""
    return _ConvFn.apply(x, weight, bias, stride, pad_lo, P, Q)


def use_salun_convs(model: nn.Module) -> int:
    """Re-class eligible nn.Conv2d modules in place.  Returns how many were switched."""
    n = 0
    for mod in model.modules():
        if hasattr(mod, "use_mfma") and hasattr(mod, "conv"):  # DDPM Downsample: asymmetric padding folded in
            mod.use_mfma = True
            n += 1
            continue
    for mod in model.modules():
        if hasattr(mod, "fused_node") and hasattr(mod, "temb_cemb_proj"):  # DDPM ResnetBlock: one autograd node
            mod.fused_node = _BLOCK_NODES[0]
    own_gemm = [m for m in model.modules() if hasattr(m, "own_gemm")]
    if own_gemm and _OWN_GEMM[0]:
        # diffusion U-Nets: attention and every Linear layer on this package's fp32 MFMA GEMM (K15, gemm.py) as well
        from .gemm import use_salun_linears
        parts = os.environ.get("SALUN_OWN_GEMM_PARTS", "linear,grouped,attn").split(",")  # A/B / debugging
        for m in own_gemm:
            is_attn = hasattr(m, "proj_out") or hasattr(m, "to_q")
            m.own_gemm = ("attn" in parts) if is_attn else ("grouped" in parts)
        if "linear" in parts:
            use_salun_linears(model)
    owners = {id(m.conv) for m in model.modules() if getattr(m, "use_mfma", False) and hasattr(m, "conv")}
    for mod in model.modules():
        if type(mod) is nn.Conv2d and _eligible(mod) and id(mod) not in owners:
            mod.__class__ = SalunConv2d
            n += 1
    # 3x3 / stride 1 / pad 1 layers: forward and backward-data on the LDS-DMA ring kernel (csrc/salun_conv_ring.hip),
    # which reads packed weight images — all layers of the model re-packed in one launch per optimizer step
    if _RING[0]:
        ringpack.register([m.weight for m in model.modules()
                           if isinstance(m, nn.Conv2d) and _eligible(m) and m.kernel_size == (3, 3)
                           and m.stride == (1, 1) and m.padding == (1, 1)])
    return n
