"""`nn.Conv2d` on the bf16 MFMA convolution kernels (csrc/salun_conv_bf16.hip, K11) — the convolution path of the
Stable-Diffusion U-Net in its bf16 configuration (BASELINE.json configs[4]; reference: autocast over the `conv_nd`
modules of SD/ldm/modules/diffusionmodules/openaimodel.py and SD/ldm/modules/attention.py:230-247).

`use_salun_convs_bf16(model)` re-classes every eligible `nn.Conv2d` in place (parameters, names, state_dict untouched:
the fp32 master weights stay views of the flat arena).  Such a module

  * takes any 4-D device tensor, views it as bf16 NHWC (`channels_last` — a no-op for tensors these modules produced),
  * reads a bf16 image of its weight that is re-packed only when the master weights changed (`ops.PARAM_EPOCH`, bumped
    by the fused optimizer kernels, plus torch's own version counter),
  * returns a bf16 `channels_last` tensor (logical NCHW shape, so the surrounding model code is unchanged),
  * in backward writes dX in bf16 and adds dW / db in fp32 straight into the parameters' `.grad` views (gradsink.py).

Shapes outside the kernels' domain (C or K not a multiple of 32: the 4-channel latent head and tail of the U-Net)
run on the fp32 MFMA kernels of conv.py; nothing here calls the library convolution.
"""
from __future__ import annotations

import os as _os_env

import torch

from .fastfn import FastFunction
import torch.nn as nn

from . import gradsink, ops
from .conv import SalunConv2d, _eligible


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """Logical NCHW tensor -> contiguous [N, H, W, C] bf16 view/copy."""
    return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


# SALUN_BF16_WGRAD_OVERLAP=0: backward-weight (+ its reduce and the bias column sums) on the main stream, as in rounds 2 - 5
_OVERLAP_BF16 = [_os_env.environ.get("SALUN_BF16_WGRAD_OVERLAP", "1") != "0"]


def _wgrad_beside(dev, tensors, launch):
    """Weight / bias gradients that go straight into `.grad` (gradsink) are consumed by nothing before the end of the
    backward pass: issue `launch()` on the side stream resblock.py keeps for exactly this (scratch buffers are per stream,
    ops.workspace; the stream has a hardware queue of its own, streams.py), next to the input-gradient kernels of this and
    the following layers — at batch 8 most of the SD U-Net's kernels leave CUs idle.  The main stream joins once, at the
    end of the backward pass (or, under data parallel, where the gradient slice is reduced: dist.BucketedGradReducer)."""
    from . import resblock
    main, side = torch.cuda.current_stream(dev.index), resblock._side_stream(dev)
    side.wait_stream(main)
    ops._STREAM_OVERRIDE[0] = side.cuda_stream  # instead of `with torch.cuda.stream(side)`: see ops._STREAM_OVERRIDE
    try:
        out = launch()
    finally:
        ops._STREAM_OVERRIDE[0] = None
    for t in tensors:
        t.record_stream(side)
    resblock.hold_until_join(tensors[-1])  # dy: autograd must not accumulate into it in place while the side stream reads it
    resblock._join_at_end_of_backward(dev)
    return out


def _can_overlap() -> bool:
    from . import resblock
    return _OVERLAP_BF16[0] and resblock.OVERLAP_WGRAD and not torch.cuda.is_current_stream_capturing()


class _ConvBF16Fn(FastFunction):
    @staticmethod
    def forward(ctx, x, w, bias, mod, nbias, addend):
        R, s, p = mod.kernel_size[0], mod.stride[0], mod.padding[0]
        xn = _nhwc(x)
        wp = mod.packed_weight()
        an = _nhwc(addend) if addend is not None else None
        y = ops.conv2d_bf16_forward(xn, wp, R, s, p, bias=bias, nbias=nbias, addend=an)
        ctx.save_for_backward(xn)  # (the weight is reached through `mod`: no saved-tensor hook for it, norm._FusedGN16)
        ctx.w_shape = tuple(w.shape)
        ctx.mod, ctx.has_bias, ctx.x_dtype = mod, bias is not None, x.dtype
        ctx.nbias, ctx.addend_dtype = nbias is not None, (addend.dtype if addend is not None else None)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xn, = ctx.saved_tensors
        mod, wshape = ctx.mod, ctx.w_shape
        R, s, p = mod.kernel_size[0], mod.stride[0], mod.padding[0]
        dyn = _nhwc(dy)
        dx = dw = db = dnb = dadd = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dst = gradsink.sink(mod.weight)  # the Parameter itself
            # the kernel has ONE accumulate flag for dw and db: without a weight sink it overwrites both outputs, so
            # the bias gradient must not target bias.grad then (it would be overwritten, not accumulated) — it goes
            # through autograd like dw; with a weight sink but no bias sink, db accumulates into fresh zeros
            bdst = gradsink.sink(mod.bias) if (ctx.has_bias and dst is not None) else None
            if ctx.has_bias and bdst is None:
                bdst = torch.zeros(wshape[0], dtype=torch.float32, device=dyn.device)
                db = bdst
            if ctx.nbias and ctx.needs_input_grad[4] and dyn.shape[0] <= 128:
                # per-image channel sums of dy out of the bias gradient's partial sums (one kernel pair for both)
                dnb = torch.empty((dyn.shape[0], wshape[0]), dtype=torch.float32, device=dyn.device)
            if dst is not None and db is None and _can_overlap():
                # everything the backward-weight call writes lands in .grad storage: side stream; dnb is read by autograd
                # on THIS stream right away, so its (small) sums are a launch of their own here
                if dnb is not None:
                    ops.colsum_bf16(dyn, dyn.shape[0], nbias_out=dnb)
                _wgrad_beside(dyn.device, (xn, dyn),
                              lambda: ops.conv2d_bf16_backward_weight(xn, dyn, wshape, s, p, out=dst, accumulate=True,
                                                                      bias_out=bdst))
            else:
                got = ops.conv2d_bf16_backward_weight(xn, dyn, wshape, s, p, out=dst, accumulate=True, bias_out=bdst,
                                                      nbias_out=dnb)
                if dst is None:
                    dw = got
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_bf16_backward_data(dyn, mod.packed_weight(), tuple(xn.shape), R, s, p).permute(0, 3, 1, 2)
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        if ctx.nbias and ctx.needs_input_grad[4] and dnb is None:  # (frozen weights: no backward-weight launch to ride on)
            dnb = dyn.float().sum(dim=(1, 2))
        if ctx.addend_dtype is not None and ctx.needs_input_grad[5]:
            dadd = dy if dy.dtype == ctx.addend_dtype else dy.to(ctx.addend_dtype)
        return dx, dw, db, None, dnb, dadd


# ------------------------------------------------------------------------------------------- weight images
# After an optimizer step EVERY bf16 weight image of a model is stale.  One pack launch per layer at its first use was
# 256 launches of ~13 us per SD step (round 5 profile: 1.4 % of the device time, most launches far too small to fill the
# chip); the modules `use_salun_convs_bf16` / `use_salun_linears_bf16` re-classed are therefore REGISTERED, and the first
# stale image any of them asks for re-packs all stale images of that device in ceil(n / 64) launches (csrc:
# k_pack_jobs).  A module that was never registered (built by hand in a test) packs alone, as before.
import weakref as _weakref

_REGISTERED: list = []      # weak references, in registration order (the order of use in a forward pass)
_IS_REGISTERED = "_salun_pack_registered"
BATCH_PACKS = [_os_env.environ.get("SALUN_BF16_BATCH_PACK", "1") != "0"]  # A/B switch: False = one launch per image at its first use
PACK_LAUNCHES = [0]


def _register(mod) -> None:
    if not getattr(mod, _IS_REGISTERED, False):
        setattr(mod, _IS_REGISTERED, True)
        _REGISTERED.append(_weakref.ref(mod))


def _weight_key(w):
    # PARAM_EPOCH: bumped by every kernel that rewrites parameters through raw pointers; w._version: torch writes
    # on the parameter itself; the flat arena's version: torch writes on `arena.params` (or any slice of it) do
    # NOT bump the parameter's own counter — `p.data = view` gave it a separate one (flat.py)
    flat = getattr(w, "_salun_flat", None)
    return (ops.PARAM_EPOCH[0], w._version, w.data_ptr(), flat._version if flat is not None else -1)


def _repack_stale(device) -> None:
    """Every registered module on `device` whose image(s) do not match its weight: one batch on the current stream."""
    jobs, done, alive = [], [], []
    for ref in _REGISTERED:
        mod = ref()
        if mod is None:
            continue
        alive.append(ref)
        w = mod.weight
        if w.device != device:
            continue
        key = _weight_key(w)
        lin = isinstance(mod, SalunLinearBF16)
        K, C = w.shape[0], w.shape[1]
        R = 1 if lin else mod.kernel_size[0]
        if mod._pack is None or mod._pack_key != key or mod._pack.device != device:
            if mod._pack is None or mod._pack.device != device:
                mod._pack = torch.empty((K, R * R, C), dtype=torch.bfloat16, device=device)
            jobs.append((w.detach(), mod._pack, K, C, R, False))
            done.append((mod, "_pack_key", key))
        if lin and (mod._pack_t is None or mod._pack_t_key != key or mod._pack_t.device != device):
            if mod._pack_t is None or mod._pack_t.device != device:
                mod._pack_t = torch.empty((C, K), dtype=torch.bfloat16, device=device)
            jobs.append((w.detach(), mod._pack_t, K, C, 1, True))
            done.append((mod, "_pack_t_key", key))
    if len(alive) != len(_REGISTERED):
        _REGISTERED[:] = alive
    PACK_LAUNCHES[0] += ops.bf16_pack_batch(jobs)
    for mod, attr, key in done:
        setattr(mod, attr, key)


class SalunConv2dBF16(nn.Conv2d):
    """Same parameters / state_dict as nn.Conv2d; device tensors go through the bf16 MFMA kernels."""

    _pack = None
    _pack_key = None

    def packed_weight(self) -> torch.Tensor:
        w = self.weight
        key = _weight_key(w)
        if self._pack is None or self._pack_key != key or self._pack.device != w.device:
            if BATCH_PACKS[0] and getattr(self, _IS_REGISTERED, False):
                _repack_stale(w.device)
            else:
                self._pack = ops.conv2d_bf16_pack(w.detach(), self._pack if self._pack is not None and self._pack.device == w.device else None)
                self._pack_key = key
                PACK_LAUNCHES[0] += 1
        return self._pack

    def forward(self, x, nbias=None, addend=None):
        """`nbias` ([N, K] fp32, e.g. a ResBlock's time-embedding term) and `addend` (a tensor of the output's shape,
        e.g. the residual branch) are added in the kernel's epilogue."""
        if not x.is_cuda or x.dim() != 4:
            raise RuntimeError("SalunConv2dBF16 needs a 4-D device tensor (the HIP kernels have no CPU path)")
        return _ConvBF16Fn.apply(x, self.weight, self.bias, self, nbias, addend)


# ----------------------------------------------------------------------------------------------- Linear layers
# A Linear layer on a token tensor [.., M tokens, C] IS a 1x1 convolution over an NHWC image whose pixels are the
# tokens — the layout the K11 kernels already read — so the transformer blocks' projections (to_q / to_k / to_v /
# to_out, the GEGLU and output projections of the feed-forward; SD/ldm/modules/attention.py:37-75,168-247) run on
# conv_bf16_igemm / conv_bf16_wgrad instead of the library GEMM: the bf16 weight image is packed once per optimizer
# step (not cast on every forward, recompute and backward), the bias (and, where the caller passes one, the residual)
# rides in the epilogue, and dW / db are added in fp32 straight into the flat gradient (no bf16 gradient, no cast, no
# AccumulateGrad launch).
def _tokens_nhwc(t: torch.Tensor, C: int) -> torch.Tensor:
    """[..., C] -> contiguous bf16 [1, H, W, C] with H * W = number of tokens (W = 8 when it divides: the backward-weight
    kernel walks 8 x 8 pixel blocks)."""
    t = t.to(torch.bfloat16).contiguous()
    M = t.numel() // C
    return t.view(1, M // 8, 8, C) if M % 8 == 0 else t.view(1, M, 1, C)


# SALUN_LINEAR_GEMM=0: keep the Linear layers on the K11 1x1 convolution kernels (A/B switch); variant pins a K16 tile
import os as _os
_USE_K16 = [_os.environ.get("SALUN_LINEAR_GEMM", "1") != "0"]
_K16_VARIANT = [int(_os.environ.get("SALUN_LINEAR_GEMM_VARIANT", "0"))]


def _al16(t) -> bool:
    return t is None or t.data_ptr() % 16 == 0


class _LinearBF16Fn(FastFunction):
    """y = x W^T + b (+ addend) on bf16 tokens.  Forward and input gradient: K16 (csrc/salun_gemm.hip, direct-to-LDS
    GEMM on the [N, K] / [K, N] weight images) when the feature counts are multiples of 64, else the K11 1x1
    convolution kernels; weight / bias gradients: K11 backward-weight, added in fp32 into the flat gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, mod, addend):
        K, C = w.shape
        x2 = x.to(torch.bfloat16).contiguous().view(-1, C)
        M = x2.shape[0]
        a2 = addend.to(torch.bfloat16).contiguous().view(-1, K) if addend is not None else None
        # K16 reads 16-byte vectors: bias / addend can be views at any 4-byte offset of the flat arena (an odd-sized
        # parameter ahead of them), in which case the K11 kernels, which take any alignment, serve the layer
        k16 = (_USE_K16[0] and ops.gemm_bf16_supported(M, K, C) and _al16(x2) and _al16(bias) and _al16(a2))
        if k16:
            y = ops.gemm_bf16_nt(x2, mod.packed_weight().view(K, C), bias, a2, _K16_VARIANT[0])
        else:
            xn = x2.view(1, M // 8, 8, C) if M % 8 == 0 else x2.view(1, M, 1, C)
            an = a2.view(1, xn.shape[1], xn.shape[2], K) if a2 is not None else None
            y = ops.conv2d_bf16_forward(xn, mod.packed_weight(), 1, 1, 0, bias=bias, nbias=None, addend=an)
        ctx.save_for_backward(x2)
        ctx.w_shape = (K, C)
        ctx.mod, ctx.has_bias, ctx.x_shape, ctx.x_dtype = mod, bias is not None, tuple(x.shape), x.dtype
        ctx.addend_dtype = addend.dtype if addend is not None else None
        return y.view(*x.shape[:-1], K)

    @staticmethod
    def backward(ctx, dy):
        x2, = ctx.saved_tensors
        mod = ctx.mod
        K, C = ctx.w_shape
        M = x2.shape[0]
        dy2 = dy.to(torch.bfloat16).contiguous().view(M, K)
        as_img = lambda t, ch: t.view(1, M // 8, 8, ch) if M % 8 == 0 else t.view(1, M, 1, ch)
        dx = dw = db = dadd = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dst = gradsink.sink(mod.weight)
            bdst = gradsink.sink(mod.bias) if (ctx.has_bias and dst is not None) else None
            if ctx.has_bias and bdst is None:
                bdst = torch.zeros(K, dtype=torch.float32, device=dy2.device)
                db = bdst
            launch = lambda: ops.conv2d_bf16_backward_weight(as_img(x2, C), as_img(dy2, K), (K, C, 1, 1), 1, 0,
                                                             out=dst.view(K, C, 1, 1) if dst is not None else None,
                                                             accumulate=True, bias_out=bdst)
            if dst is not None and db is None and _can_overlap():
                _wgrad_beside(dy2.device, (x2, dy2), launch)
            else:
                got = launch()
                if dst is None:
                    dw = got.view(K, C)
        if ctx.needs_input_grad[0]:
            if _USE_K16[0] and ops.gemm_bf16_supported(M, C, K) and _al16(dy2):
                dx = ops.gemm_bf16_nt(dy2, mod.packed_weight_t(), None, None, _K16_VARIANT[0]).view(ctx.x_shape)
            else:
                dx = ops.conv2d_bf16_backward_data(as_img(dy2, K), mod.packed_weight(), (1,) + tuple(as_img(x2, C).shape[1:]),
                                                   1, 1, 0).view(ctx.x_shape)
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        if ctx.addend_dtype is not None and ctx.needs_input_grad[4]:
            dadd = dy if dy.dtype == ctx.addend_dtype else dy.to(ctx.addend_dtype)
        return dx, dw, db, None, dadd


class SalunLinearBF16(nn.Linear):
    """Same parameters / state_dict as nn.Linear.  Device inputs in the bf16 configuration (a bf16 tensor, or any
    tensor under bf16 autocast) go through the bf16 MFMA kernels; anything else is the plain fp32 F.linear."""

    _pack = None
    _pack_key = None
    _pack_t = None
    _pack_t_key = None

    def _key(self):
        return _weight_key(self.weight)

    def packed_weight(self) -> torch.Tensor:
        """bf16 image [N, 1, K] (= [N, K]) of the master weights, re-packed once per optimizer step."""
        w = self.weight
        key = self._key()
        if self._pack is None or self._pack_key != key or self._pack.device != w.device:
            if BATCH_PACKS[0] and getattr(self, _IS_REGISTERED, False):
                _repack_stale(w.device)
            else:
                K, C = w.shape
                self._pack = ops.conv2d_bf16_pack(w.detach().view(K, C, 1, 1),
                                                  self._pack if self._pack is not None and self._pack.device == w.device else None)
                self._pack_key = key
                PACK_LAUNCHES[0] += 1
        return self._pack

    def packed_weight_t(self) -> torch.Tensor:
        """bf16 image [K, N] (the transposed weights the input-gradient GEMM reads)."""
        w = self.weight
        key = self._key()
        if self._pack_t is None or self._pack_t_key != key or self._pack_t.device != w.device:
            if BATCH_PACKS[0] and getattr(self, _IS_REGISTERED, False):
                _repack_stale(w.device)
            else:
                self._pack_t = ops.pack_bf16(w.detach(), True,
                                             self._pack_t if self._pack_t is not None and self._pack_t.device == w.device else None)
                self._pack_t_key = key
                PACK_LAUNCHES[0] += 1
        return self._pack_t

    def forward(self, x, addend=None):
        """`addend` (a tensor of the output's shape, e.g. the residual branch) is added in the kernel's epilogue."""
        bf16_mode = x.is_cuda and (x.dtype == torch.bfloat16 or
                                   (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16))
        if bf16_mode:
            return _LinearBF16Fn.apply(x, self.weight, self.bias, self, addend)
        y = torch.nn.functional.linear(x, self.weight, self.bias)
        return y if addend is None else y + addend


def use_salun_linears_bf16(model: nn.Module) -> int:
    """Re-class the nn.Linear layers of the transformer blocks (feature counts that are multiples of 32) in place;
    returns how many were switched.  The two tiny time-embedding Linears (M = batch rows) stay on the library."""
    from .SD.unet import BasicTransformerBlock
    n = 0
    for blk in model.modules():
        if not isinstance(blk, BasicTransformerBlock):
            continue
        for mod in blk.modules():
            if type(mod) is nn.Linear and mod.in_features % 32 == 0 and mod.out_features % 32 == 0 and \
                    ops.conv2d_bf16_supported(mod.in_features, mod.out_features, 1, 1, 0):
                mod.__class__ = SalunLinearBF16
                _register(mod)
                n += 1
    return n


class _Fp32Island(SalunConv2d):
    """The two 4-channel convolutions of the U-Net (latent head / tail) inside a bf16 model: fp32 MFMA kernels on an
    fp32 NCHW copy of the input; the result keeps fp32 (the head feeds GroupNorm, the tail is the model output)."""

    def forward(self, x):
        with torch.autocast("cuda", enabled=False):  # SalunConv2d refuses autocast regions; this island is fp32 by design
            y = super().forward(x.float().contiguous())
        if self.out_channels % 32 == 0:  # the head: its output feeds the bf16 NHWC network (and the skip concatenations)
            y = y.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        return y


def use_salun_convs_bf16(model: nn.Module) -> int:
    """Re-class eligible nn.Conv2d modules in place; returns how many run on the bf16 kernels."""
    n = 0
    for mod in model.modules():
        if type(mod) is not nn.Conv2d or not _eligible(mod):
            continue
        K, C = mod.out_channels, mod.in_channels
        R, s, p = mod.kernel_size[0], mod.stride[0], mod.padding[0]
        if C % 32 == 0 and K % 32 == 0 and ops.conv2d_bf16_supported(C, K, R, s, p):
            mod.__class__ = SalunConv2dBF16
            _register(mod)
            n += 1
        else:
            mod.__class__ = _Fp32Island
    return n
