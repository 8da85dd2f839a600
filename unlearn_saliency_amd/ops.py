"""Tensor-level wrappers over the C-ABI (include/salun.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function
below hands raw device pointers + the current stream to libsalun.so.  All tensors must
live on a ROCm device (``tensor.is_cuda``); there is deliberately no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from .fastfn import FastFunction

from . import _lib, ringpack
from ._lib import c_double, c_int, c_int64, c_size_t, c_uint64, c_void_p, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


# Set (to a raw stream handle) around calls that must be issued on another stream than torch's current one WITHOUT paying
# for `with torch.cuda.stream(...)` (~10 us of host time per use; conv_bf16._wgrad_beside issues 470 backward-weight calls
# per SD step on the side stream and the step is host-bound).  Only the wrappers of this module look at it: a wrapper
# that allocates its result with torch must not be called under an override (the allocation would belong to the current
# stream) — backward-weight into `.grad` storage allocates nothing but the per-stream scratch buffer.
_STREAM_OVERRIDE = [None]


def _stream_handle(idx: Optional[int] = None) -> int:
    """The raw handle of device `idx`'s (default: the current device's) current stream — what
    `torch.cuda.current_stream(idx).cuda_stream` returns, without building a Stream object and re-probing the device on
    each of ~4,500 calls per SD step."""
    if _STREAM_OVERRIDE[0] is not None:
        return _STREAM_OVERRIDE[0]
    if _raw_stream is None or _cur_device is None:
        return torch.cuda.current_stream(idx).cuda_stream
    return _raw_stream(_cur_device() if idx is None else idx)


def _stream() -> c_void_p:
    return c_void_p(_stream_handle())


def _dev(t: Optional[torch.Tensor], dtype: torch.dtype, name: str, allow_none: bool = False) -> c_void_p:
    if t is None:
        if allow_none:
            return c_void_p(None)
        raise ValueError(f"{name} must not be None")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the SalUn HIP kernels need a device tensor (got {t.device}); no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return c_void_p(t.data_ptr())


# Shape-only queries of the library (workspace sizes, "is this shape in the kernel's domain") memoised on the host: each is
# a ctypes round trip, and the SD step — host-bound since round 6 — asked ~3,000 of them per step with a few dozen
# distinct argument tuples.
_shape_memo: dict = {}


def _q(name: str, *args):
    key = (name,) + args
    v = _shape_memo.get(key)
    if v is None:
        fn = getattr(_lib.lib(), name)
        v = _shape_memo[key] = fn(*[c_int64(a) if t is c_int64 else a for a, t in zip(args, fn.argtypes)])
    return v


# One scratch buffer per (device, stream), grown on demand: reuse is ordered by the stream the kernels run on, so
# work issued on a side stream (resblock.py overlaps backward-weight with backward-data) never shares scratch with
# the main stream.
_ws: dict[tuple, torch.Tensor] = {}


def workspace(nbytes: int, device: torch.device, tag: str = "") -> torch.Tensor:
    """`tag` separates buffers whose contents must survive other ops' scratch use (the top-k publication block is
    read back by mask_topk_thresholds after arbitrary calls in between)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    # the top-k publication block is keyed by the device alone: mask_topk_status / mask_topk_thresholds must find the
    # block the last mask_topk call wrote whatever stream is current when they are called (side streams exist)
    key = (idx, None if tag == "topk" else _stream_handle(idx), tag)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


# Bumped by every kernel that rewrites parameters through raw pointers (the fused optimizer steps, the proximal
# step): torch's version counters do not see those writes, and derived images of the weights (the bf16 weight pack of
# conv_bf16.py) are cached against this counter.
PARAM_EPOCH = [0]


# ----------------------------------------------------------------------------- K1
def saliency_accumulate(acc: torch.Tensor, g: torch.Tensor, scale: float = 1.0,
                        sqnorm: Optional[torch.Tensor] = None, max_norm: float = 1.0) -> None:
    """acc += g*scale  (or g * clip_coef(sqnorm, max_norm) when `sqnorm` is given)."""
    assert acc.numel() == g.numel()
    check(_lib.lib().salun_saliency_accumulate(_dev(acc, torch.float32, "acc"), _dev(g, torch.float32, "g"),
                                               c_double(scale), _dev(sqnorm, torch.float32, "sqnorm", True),
                                               c_double(max_norm), c_int64(acc.numel()), _stream()),
          "salun_saliency_accumulate")


# ----------------------------------------------------------------------------- K2
class TopkFailed(RuntimeError):
    """The full-scan route's grid barrier timed out (its workgroups were not co-resident): the masks are garbage."""


def mask_topk(acc: torch.Tensor, ks: Sequence[int], out: Optional[Sequence[torch.Tensor]] = None,
              flags: int = 0, check: bool = False) -> list[torch.Tensor]:
    """One u8 0/1 mask per k: the k largest |acc| (ties: lowest flat index first).
    `flags`: _lib.SALUN_TOPK_* (FORCE_FULL_SCAN for A/B tests, VALUES_ONLY publishes the thresholds without
    writing masks — returns []).
    `check=True` reads the device-side status after the call (ONE host sync) and raises `TopkFailed` instead of
    returning invalid masks — what every caller that hands masks to a file or an optimizer does (mask generation is
    followed by a synchronising save anyway).  Callers that must not synchronise (the proximal step) get the failure
    through the exported threshold instead: it is NaN, and `salun_soft_threshold_step` poisons the weights with it."""
    L = _lib.lib()
    n, nk = acc.numel(), len(ks)
    if not 1 <= nk <= _lib.SALUN_MAX_THRESHOLDS:
        raise ValueError(f"1 <= len(ks) <= {_lib.SALUN_MAX_THRESHOLDS}")
    values_only = bool(flags & _lib.SALUN_TOPK_VALUES_ONLY)
    if values_only:
        out, marr = [], None
    else:
        if out is None:  # one allocation for all thresholds: the masks are rows of it, each row 256-byte aligned (the
            # single-read route stores the masks as dwords).  NOTE: the nk masks SHARE that storage — keeping
            # one keeps all nk * n bytes resident, and pickling a row serialises the whole block; callers that keep a
            # subset for long (the unlearning loops keep one ratio) should `.clone()` it, the save paths already go
            # through unpack_mask, which copies
            n_pad = (n + 255) & ~255
            out = [r[:n] for r in torch.empty((nk, n_pad), dtype=torch.uint8, device=acc.device).unbind(0)]
        assert len(out) == nk and all(o.numel() == n for o in out)
        marr = (c_void_p * nk)(*[_dev(o, torch.uint8, "mask").value for o in out])
    nbytes = L.salun_mask_topk_workspace_bytes(c_int64(n), c_int(nk))
    ws = workspace(nbytes, acc.device, "topk")
    karr = (c_int64 * nk)(*[int(k) for k in ks])
    _lib.check(L.salun_mask_topk_ex(_dev(acc, torch.float32, "acc"), c_int64(n), karr, c_int(nk), marr,
                                    c_void_p(ws.data_ptr()), c_size_t(ws.numel()), ctypes.c_uint(flags), _stream()),
               "salun_mask_topk")  # (`check` the keyword shadows the module-level helper inside this function)
    if check:
        route, err = mask_topk_status(acc.device)
        from . import dist as sdist
        if sdist.collectives_on():
            # every rank ranks the same vector, but a time-out is a property of one device's occupancy: agree on the
            # flag before raising, or the healthy ranks hang at their next collective while one rank unwinds
            flag = torch.tensor([int(err)], dtype=torch.int32, device=acc.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            err = int(flag.item())
        if err:
            raise TopkFailed(f"salun_mask_topk: the full scan's grid barrier timed out (route {route}, n = {n}): "
                             "its workgroups were not co-resident; no mask was produced")
    return list(out)


def mask_topk_status(device: torch.device) -> tuple[int, int]:
    """(route, error) of the LAST mask_topk call on this device: route 1 = single-read, 2 = full scan; error 1 = a grid
    barrier of the full scan timed out.  Synchronises the stream (diagnostics / tests only)."""
    ws = workspace(0, device, "topk")
    route, err = c_int(0), c_int(0)
    check(_lib.lib().salun_mask_topk_status(c_void_p(ws.data_ptr()), ctypes.byref(route), ctypes.byref(err), _stream()),
          "salun_mask_topk_status")
    return route.value, err.value


def mask_topk_thresholds(device: torch.device, nk: int) -> torch.Tensor:
    """|acc| value of the k-th element for the thresholds of the LAST mask_topk call on this device."""
    ws = workspace(0, device, "topk")
    out = torch.empty(nk, dtype=torch.float32, device=device)
    check(_lib.lib().salun_mask_topk_thresholds(c_void_p(ws.data_ptr()), c_int(nk), c_void_p(out.data_ptr()),
                                                _stream()), "salun_mask_topk_thresholds")
    return out


def mask_u8_to_i64(m: torch.Tensor) -> torch.Tensor:
    out = torch.empty(m.shape, dtype=torch.int64, device=m.device)
    check(_lib.lib().salun_mask_u8_to_i64(_dev(m, torch.uint8, "mask"), c_void_p(out.data_ptr()),
                                          c_int64(m.numel()), _stream()), "salun_mask_u8_to_i64")
    return out


def mask_i64_to_u8(m: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(m.shape, dtype=torch.uint8, device=m.device)
    check(_lib.lib().salun_mask_i64_to_u8(_dev(m, torch.int64, "mask"), _dev(out, torch.uint8, "out"),
                                          c_int64(m.numel()), _stream()), "salun_mask_i64_to_u8")
    return out


def mask_popcount(m: torch.Tensor) -> int:
    L = _lib.lib()
    ws = workspace(L.salun_reduce_workspace_bytes(c_int64(m.numel())), m.device)
    out = torch.empty(1, dtype=torch.int64, device=m.device)
    check(L.salun_mask_popcount(_dev(m, torch.uint8, "mask"), c_int64(m.numel()), c_void_p(out.data_ptr()),
                                c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()), "salun_mask_popcount")
    return int(out.item())


# -------------------------------------------------------------------------- K3+K4
def masked_sgd_step(p: torch.Tensor, g: torch.Tensor, buf: Optional[torch.Tensor], m: Optional[torch.Tensor],
                    lr: float, momentum: float, weight_decay: float, first_step: bool) -> None:
    n = p.numel()
    assert g.numel() == n and (buf is None or buf.numel() == n) and (m is None or m.numel() == n)
    PARAM_EPOCH[0] += 1
    check(_lib.lib().salun_masked_sgd_step(_dev(p, torch.float32, "p"), _dev(g, torch.float32, "g"),
                                           _dev(buf, torch.float32, "buf", True), _dev(m, torch.uint8, "mask", True),
                                           c_double(lr), c_double(momentum), c_double(weight_decay),
                                           c_int(int(first_step)), c_int64(n), _stream()), "salun_masked_sgd_step")


# ----------------------------------------------------------------------------- K5
def grad_sqnorm(g: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum(g^2) as a 1-element device tensor (no host sync)."""
    L = _lib.lib()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=g.device)
    ws = workspace(L.salun_reduce_workspace_bytes(c_int64(g.numel())), g.device)
    check(L.salun_grad_sqnorm(_dev(g, torch.float32, "g"), c_int64(g.numel()), _dev(out, torch.float32, "out"),
                              c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()), "salun_grad_sqnorm")
    return out


def masked_adam_step(p: torch.Tensor, g: torch.Tensor, m1: torch.Tensor, v: torch.Tensor,
                     mask: Optional[torch.Tensor], lr: float, beta1: float, beta2: float, eps: float,
                     weight_decay: float, step: int, sqnorm: Optional[torch.Tensor] = None,
                     max_norm: float = 1.0, gscale: float = 1.0) -> None:
    n = p.numel()
    assert g.numel() == n and m1.numel() == n and v.numel() == n and (mask is None or mask.numel() == n)
    PARAM_EPOCH[0] += 1
    check(_lib.lib().salun_masked_adam_step(_dev(p, torch.float32, "p"), _dev(g, torch.float32, "g"),
                                            _dev(m1, torch.float32, "exp_avg"), _dev(v, torch.float32, "exp_avg_sq"),
                                            _dev(mask, torch.uint8, "mask", True),
                                            _dev(sqnorm, torch.float32, "sqnorm", True), c_double(max_norm),
                                            c_double(gscale), c_double(lr), c_double(beta1), c_double(beta2),
                                            c_double(eps), c_double(weight_decay), c_int(int(step)), c_int64(n),
                                            _stream()), "salun_masked_adam_step")


def adam_coefficients(step_dev: torch.Tensor, lr: float, beta1: float, beta2: float, coef: torch.Tensor) -> None:
    """++step_dev (int64[1], device) and coef <- {sqrt(1 - b2^t), -lr / (1 - b1^t)} on the stream."""
    check(_lib.lib().salun_adam_coefficients(_dev(step_dev, torch.int64, "step"), c_double(lr), c_double(beta1),
                                             c_double(beta2), _dev(coef, torch.float32, "coef"), _stream()),
          "salun_adam_coefficients")


def masked_adam_step_coef(p, g, m1, v, mask, coef, beta1, beta2, eps, weight_decay, sqnorm=None, max_norm=1.0,
                          gscale=1.0) -> None:
    """masked_adam_step with the step-dependent scalars read from device memory (whole-step HIP graphs)."""
    n = p.numel()
    PARAM_EPOCH[0] += 1
    check(_lib.lib().salun_masked_adam_step_coef(_dev(p, torch.float32, "p"), _dev(g, torch.float32, "g"),
                                                 _dev(m1, torch.float32, "exp_avg"), _dev(v, torch.float32, "exp_avg_sq"),
                                                 _dev(mask, torch.uint8, "mask", True),
                                                 _dev(sqnorm, torch.float32, "sqnorm", True), c_double(max_norm),
                                                 c_double(gscale), _dev(coef, torch.float32, "coef"), c_double(beta1),
                                                 c_double(beta2), c_double(eps), c_double(weight_decay), c_int64(n),
                                                 _stream()), "salun_masked_adam_step_coef")


# ----------------------------------------------------------------------------- K6
def qsample(x0: torch.Tensor, e: torch.Tensor, sqrt_ab: torch.Tensor, sqrt_1mab: torch.Tensor,
            t: torch.Tensor) -> torch.Tensor:
    B = x0.shape[0]
    chw = x0.numel() // max(B, 1)
    xt = torch.empty_like(x0)
    check(_lib.lib().salun_qsample(_dev(x0, torch.float32, "x0"), _dev(e, torch.float32, "e"),
                                   _dev(sqrt_ab, torch.float32, "sqrt_ab"), _dev(sqrt_1mab, torch.float32, "sqrt_1mab"),
                                   _dev(t, torch.int64, "t"), c_int64(sqrt_ab.numel()), c_void_p(xt.data_ptr()),
                                   c_int64(B), c_int64(chw), _stream()), "salun_qsample")
    return xt


def sqerr_loss(a: torch.Tensor, b: torch.Tensor, coef: float, want_per_sample: bool = False,
               want_grad: bool = True):
    """loss = coef * sum((a-b)^2); returns (loss[1], per_sample[B] or None, dloss/db or None)."""
    L = _lib.lib()
    B = a.shape[0]
    chw = a.numel() // B
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    per = torch.empty(B, dtype=torch.float32, device=a.device) if want_per_sample else None
    d = torch.empty_like(b) if want_grad else None
    ws = workspace(L.salun_sqerr_workspace_bytes(c_int64(B), c_int64(chw)), a.device)
    check(L.salun_sqerr_loss(_dev(a, torch.float32, "a"), _dev(b, torch.float32, "b"), c_int64(B), c_int64(chw),
                             c_double(coef), c_void_p(loss.data_ptr()), _dev(per, torch.float32, "per_sample", True),
                             _dev(d, torch.float32, "dloss", True), c_void_p(ws.data_ptr()), c_size_t(ws.numel()),
                             _stream()), "salun_sqerr_loss")
    return loss, per, d


class _SqErr(FastFunction):
    """coef * sum((target - pred)^2) with the gradient produced in the same pass."""

    @staticmethod
    def forward(ctx, target, pred, coef):
        loss, _, d = sqerr_loss(target.contiguous(), pred.contiguous(), coef, want_grad=True)
        ctx.save_for_backward(d)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        (d,) = ctx.saved_tensors
        return None, d * grad_out, None


def eps_mse(e: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """(e - out).square().sum(dim=(1,2,3)).mean(dim=0) — DDPM/functions/losses.py:37."""
    return _SqErr.apply(e, out, 1.0 / e.shape[0])


def mse_loss(target: torch.Tensor, pred: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss()(target, pred) with grad flowing to `pred` — DDPM/runners/diffusion.py:570."""
    return _SqErr.apply(target, pred, 1.0 / target.numel())


def dropout(x: torch.Tensor, p: float, key: int, sample_offset: int = 0, out: Optional[torch.Tensor] = None,
            seed_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Counter-based dropout of a (n, ...) fp32 batch whose first sample is global sample `sample_offset`
    (salun_dropout); applying it to dy with the same (key, offset) is the backward pass.  `out` may be `x`."""
    n = x.shape[0]
    chw = x.numel() // max(n, 1)
    y = out if out is not None else torch.empty_like(x)
    check(_lib.lib().salun_dropout(_dev(x, torch.float32, "x"), _dev(y, torch.float32, "y"), c_int64(n), c_int64(chw),
                                   c_int64(sample_offset), c_double(p), c_uint64(key & 0xFFFFFFFFFFFFFFFF),
                                   _dev(seed_dev, torch.int64, "seed_dev", True), _stream()), "salun_dropout")
    return y


class _Dropout(FastFunction):
    @staticmethod
    def forward(ctx, x, p, key, offset):
        ctx.cfg = (float(p), int(key), int(offset))
        return dropout(x.contiguous(), p, key, offset)

    @staticmethod
    def backward(ctx, dy):
        p, key, offset = ctx.cfg
        return dropout(dy.contiguous(), p, key, offset), None, None, None


def dropout_fn(x: torch.Tensor, p: float, key: int, sample_offset: int = 0) -> torch.Tensor:
    """Differentiable counter-based dropout (no mask tensor is kept: backward re-derives it from the key)."""
    return _Dropout.apply(x, float(p), int(key), int(sample_offset))


# ----------------------------------------------------------------------------- K7
def fim_square_accumulate(F: torch.Tensor, tmp: torch.Tensor, n_data: float) -> None:
    check(_lib.lib().salun_fim_square_accumulate(_dev(F, torch.float32, "F"), _dev(tmp, torch.float32, "tmp"),
                                                 c_double(n_data), c_int64(F.numel()), _stream()),
          "salun_fim_square_accumulate")


# ----------------------------------------------------------------------------- K8
_data_ws_cache: dict = {}


def _data_ws_bytes(N: int, outC: int, outH: int, outW: int, R: int, conv_stride: int) -> int:
    """salun_conv2d_data_workspace_bytes, remembered per shape (it is asked for every convolution call)."""
    key = (N, outC, outH, outW, R, conv_stride)
    v = _data_ws_cache.get(key)
    if v is None:
        v = _data_ws_cache[key] = int(_lib.lib().salun_conv2d_data_workspace_bytes(N, outC, outH, outW, R, conv_stride))
    return v


def conv2d_forward(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int, pad: int,
                   P: int, Q: int, nbias: Optional[torch.Tensor] = None,
                   addend: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """y = conv2d(x, w) (+ bias[k]) (+ nbias[n, k]) (+ addend[n, k, p, q]) on the matrix cores; None if the shape is
    outside the kernel's tiling domain."""
    N, C, H, W = x.shape
    K, _, R, _ = w.shape
    if nbias is not None and tuple(nbias.shape) != (N, K):
        raise ValueError(f"nbias must be [{N}, {K}], got {tuple(nbias.shape)}")
    if addend is not None and tuple(addend.shape) != (N, K, P, Q):
        raise ValueError(f"addend must be {(N, K, P, Q)}, got {tuple(addend.shape)}")
    if R == 3 and stride == 1 and pad == 1 and P == H and Q == W:
        imgs = ringpack.images(w)  # registered weights (conv.use_salun_convs): the LDS-DMA ring kernel, K8r
        if imgs is not None:
            y = conv3x3_packed(x, imgs[0], K, bias=bias, nbias=nbias, addend=addend)
            if y is not None:
                return y
    y = torch.empty((N, K, P, Q), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    wsb = _data_ws_bytes(N, K, P, Q, R, stride)  # > 0 only for under-filled launches
    ws = workspace(wsb, x.device) if wsb else None
    rc = L.salun_conv2d_forward_fused(_dev(x, torch.float32, "x"), _dev(w, torch.float32, "w"),
                                      _dev(bias, torch.float32, "bias", True),
                                      _dev(nbias, torch.float32, "nbias", True),
                                      _dev(addend, torch.float32, "addend", True), c_void_p(y.data_ptr()),
                                      N, C, H, W, K, R, stride, pad, P, Q,
                                      c_void_p(ws.data_ptr() if ws is not None else None), c_size_t(wsb), _stream())
    if rc == _lib.SALUN_EINVAL:
        return None
    check(rc, "salun_conv2d_forward_fused")
    return y


def conv2d_backward_data(dy: torch.Tensor, w: torch.Tensor, x_shape, stride: int, pad: int,
                         addend: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """dx = backward_data(dy, w) (+ addend, folded into the epilogue); None outside the tiling domain."""
    N, C, H, W = x_shape
    K, _, R, _ = w.shape
    P, Q = dy.shape[2], dy.shape[3]
    if R == 3 and stride == 1 and pad == 1 and P == H and Q == W:
        imgs = ringpack.images(w)
        if imgs is not None:
            dx = conv3x3_packed(dy, imgs[1], C, addend=addend)
            if dx is not None:
                return dx
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    L = _lib.lib()
    wsb = _data_ws_bytes(N, C, H, W, R, 1) if stride == 1 else 0
    ws = workspace(wsb, dy.device) if wsb else None
    rc = L.salun_conv2d_backward_data_ws(_dev(dy, torch.float32, "dy"), _dev(w, torch.float32, "w"),
                                         _dev(addend, torch.float32, "addend", True),
                                         c_void_p(dx.data_ptr()), N, C, H, W, K, R, stride, pad, P, Q,
                                         c_void_p(ws.data_ptr() if ws is not None else None), c_size_t(wsb), _stream())
    if rc == _lib.SALUN_EINVAL:
        return None
    check(rc, "salun_conv2d_backward_data")
    return dx


# None: the caller's `shared` flag decides the 3x3 / stride 1 backward-weight kernel; "shared" / "alone" pin one (tests that
# compare a side-stream schedule with the single-stream one bit for bit need the same kernel in both)
WGRAD_KERNEL = [None]


def conv2d_backward_weight(x: torch.Tensor, dy: torch.Tensor, w_shape, stride: int, pad: int,
                           out: Optional[torch.Tensor] = None, accumulate: bool = False,
                           shared: bool = False) -> Optional[torch.Tensor]:
    """dw = backward_weight(x, dy); with `out` the result is written (accumulate=False) or added
    (accumulate=True) into that tensor — e.g. the parameter's slice of the flat gradient arena.  `shared`: the launch
    runs on a side stream beside other kernels of the step (SALUN_WGRAD_SHARED: kernel choice, include/salun.h)."""
    N, C, H, W = x.shape
    K, _, R, _ = w_shape
    P, Q = dy.shape[2], dy.shape[3]
    L = _lib.lib()
    nbytes = L.salun_conv2d_wgrad_workspace_bytes(N, C, K, R, P, Q)
    if nbytes == 0:
        return None
    ws = workspace(nbytes, x.device)
    dw = out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    rc = L.salun_conv2d_backward_weight_ex(_dev(x, torch.float32, "x"), _dev(dy, torch.float32, "dy"),
                                           _dev(dw, torch.float32, "dw"), N, C, H, W, K, R, stride, pad, P, Q,
                                           int(bool(accumulate and out is not None)),
                                           _lib.SALUN_WGRAD_SHARED if (shared if WGRAD_KERNEL[0] is None
                                                                       else WGRAD_KERNEL[0] == "shared") else 0,
                                           c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream())
    if rc == _lib.SALUN_EINVAL:
        return None
    check(rc, "salun_conv2d_backward_weight")
    return dw


# ----------------------------------------------------------------------------- K8r
# 3x3 / stride 1 / pad 1 convolution on the LDS-DMA ring kernel (csrc/salun_conv_ring.hip): packed weight images.
def conv3x3_pack(w: torch.Tensor, dgrad: bool, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Packed image of an OIHW [K, C, 3, 3] weight for the forward (dgrad=False) or backward-data (dgrad=True) walk;
    None when the reduction channel count is not a multiple of 8 (the caller keeps the conv_igemm path).  (Models pack
    all their layers in one launch through ringpack.py; this is the single-image form for tools and tests.)"""
    K, C = w.shape[0], w.shape[1]
    L = _lib.lib()
    nbytes = int(L.salun_conv3x3_pack_bytes(K, C, int(dgrad)))
    if nbytes == 0 or tuple(w.shape[2:]) != (3, 3):
        return None
    img = out if out is not None and out.numel() * 4 == nbytes else torch.empty(nbytes // 4, dtype=torch.float32,
                                                                                device=w.device)
    _dev(w, torch.float32, "w")
    job = _lib.PackJob(w.data_ptr(), None if dgrad else img.data_ptr(), img.data_ptr() if dgrad else None, K, C)
    check(L.salun_conv3x3_pack_weights(ctypes.cast(ctypes.pointer(job), c_void_p), 1, _stream()),
          "salun_conv3x3_pack_weights")
    return img


def conv3x3_packed(x: torch.Tensor, img: torch.Tensor, Kout: int, bias: Optional[torch.Tensor] = None,
                   nbias: Optional[torch.Tensor] = None, addend: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, cfg: int = 0) -> Optional[torch.Tensor]:
    """y[N, Kout, H, W] = conv3x3(x[N, Cred, H, W], packed image) (+ bias) (+ nbias[n, k]) (+ addend); None outside the
    kernel's tiling domain.  `out` may be the addend (accumulate in place)."""
    N, Cred, H, W = x.shape
    y = out if out is not None else torch.empty((N, Kout, H, W), dtype=torch.float32, device=x.device)
    rc = _lib.lib().salun_conv3x3_packed(_dev(x, torch.float32, "x"), c_void_p(img.data_ptr()),
                                         _dev(bias, torch.float32, "bias", True),
                                         _dev(nbias, torch.float32, "nbias", True),
                                         _dev(addend, torch.float32, "addend", True), c_void_p(y.data_ptr()),
                                         N, Cred, H, W, Kout, int(cfg), _stream())
    if rc == _lib.SALUN_EINVAL:
        return None
    check(rc, "salun_conv3x3_packed")
    return y


def channel_sum(dy: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """Bias gradient of a convolution: sum of dy[N, K, P, Q] over n, p, q -> [K] (written, or added into `out`)."""
    N, K = dy.shape[0], dy.shape[1]
    HW = dy.shape[2] * dy.shape[3]
    L = _lib.lib()
    ws = workspace(L.salun_channel_sum_workspace_bytes(N, K), dy.device)
    res = out if out is not None else torch.empty(K, dtype=torch.float32, device=dy.device)
    check(L.salun_channel_sum(_dev(dy, torch.float32, "dy"), _dev(res, torch.float32, "out"), N, K, HW,
                              int(bool(accumulate and out is not None)), c_void_p(ws.data_ptr()), c_size_t(ws.numel()),
                              _stream()), "salun_channel_sum")
    return res


# ----------------------------------------------------------------------------- K11
# bf16 NHWC convolution on the bf16 matrix-core instruction; tensors are [N, H, W, C] contiguous bfloat16.
def conv2d_bf16_supported(C: int, K: int, R: int, stride: int, pad: int) -> bool:
    return bool(_q("salun_conv2d_bf16_supported", C, K, R, stride, pad))


# number of weight re-packs issued so far (SD/train_scripts.py::forget_and_target uses it to detect cold caches: a pack
# kernel enqueued on one stream while another stream is about to read the same image)
PACK_CALLS = [0]


def conv2d_bf16_pack(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 OIHW master weights -> the bf16 image [K, R*R, C] the forward and backward-data kernels read."""
    K, C, R, _ = w.shape
    PACK_CALLS[0] += 1
    if out is None:
        out = torch.empty((K, R * R, C), dtype=torch.bfloat16, device=w.device)
    check(_lib.lib().salun_conv2d_bf16_pack_weights(_dev(w, torch.float32, "w"), _dev(out, torch.bfloat16, "wp"), K, C, R,
                                                    _stream()), "salun_conv2d_bf16_pack_weights")
    return out


def bf16_pack_batch(jobs) -> int:
    """jobs: iterable of (w fp32 OIHW / [K, C], wp bf16 image, K, C, R, transposed) -> every image written by
    ceil(n / 64) launches on the current stream (the per-model form of conv2d_bf16_pack / pack_bf16; conv_bf16.py keeps the
    registry).  Returns the number of launches."""
    jobs = list(jobs)
    if not jobs:
        return 0
    PACK_CALLS[0] += 1
    L, st, cap = _lib.lib(), _stream(), _lib.SALUN_BF16_PACK_MAX_JOBS
    launches = 0
    for i in range(0, len(jobs), cap):
        part = jobs[i:i + cap]
        arr = (_lib.Bf16PackJob * len(part))()
        for a, (w, wp, K, C, R, tr) in zip(arr, part):
            _dev(w, torch.float32, "w")
            _dev(wp, torch.bfloat16, "wp")
            a.w, a.wp, a.K, a.C, a.R, a.transposed = w.data_ptr(), wp.data_ptr(), K, C, R, int(bool(tr))
        check(L.salun_bf16_pack_weights_batch(ctypes.cast(arr, c_void_p), len(part), st), "salun_bf16_pack_weights_batch")
        launches += 1
    return launches


def conv2d_bf16_forward(x: torch.Tensor, wp: torch.Tensor, R: int, stride: int, pad: int,
                        bias: Optional[torch.Tensor] = None, nbias: Optional[torch.Tensor] = None,
                        addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    N, H, W, C = x.shape
    K = wp.shape[0]
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    _dev(x, torch.bfloat16, "x")  # device / dtype / layout errors before anything touches the device
    y = torch.empty((N, OH, OW, K), dtype=torch.bfloat16, device=x.device)
    ws = workspace(_q("salun_conv2d_bf16_data_workspace_bytes", N, H, W, C, K, R, stride, pad), x.device)
    check(_lib.lib().salun_conv2d_bf16_forward(_dev(x, torch.bfloat16, "x"), _dev(wp, torch.bfloat16, "wp"),
                                               _dev(bias, torch.float32, "bias", True),
                                               _dev(nbias, torch.float32, "nbias", True),
                                               _dev(addend, torch.bfloat16, "addend", True), c_void_p(y.data_ptr()),
                                               N, H, W, C, K, R, stride, pad, c_void_p(ws.data_ptr()),
                                               c_size_t(ws.numel()), _stream()), "salun_conv2d_bf16_forward")
    return y


def conv2d_bf16_backward_data(dy: torch.Tensor, wp: torch.Tensor, x_shape, R: int, stride: int, pad: int,
                              addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    N, H, W, C = x_shape
    K = wp.shape[0]
    _dev(dy, torch.bfloat16, "dy")
    dx = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=dy.device)
    ws = workspace(_q("salun_conv2d_bf16_data_workspace_bytes", N, H, W, C, K, R, stride, pad), dy.device)
    check(_lib.lib().salun_conv2d_bf16_backward_data(_dev(dy, torch.bfloat16, "dy"), _dev(wp, torch.bfloat16, "wp"),
                                                     _dev(addend, torch.bfloat16, "addend", True),
                                                     c_void_p(dx.data_ptr()), N, H, W, C, K, R, stride, pad,
                                                     c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()),
          "salun_conv2d_bf16_backward_data")
    return dx


def conv2d_bf16_backward_weight(x: torch.Tensor, dy: torch.Tensor, w_shape, stride: int, pad: int,
                                out: Optional[torch.Tensor] = None, accumulate: bool = False,
                                bias_out: Optional[torch.Tensor] = None,
                                nbias_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw fp32 OIHW (written, or added into `out` when accumulate); `bias_out` (fp32 [K]) receives / accumulates the
    per-channel sum of dy in the same call; `nbias_out` (fp32 [N, K], N <= 128) is overwritten with the per-image channel
    sums of dy — the gradient of the forward's `nbias` term."""
    N, H, W, C = x.shape
    K, _, R, _ = w_shape
    L = _lib.lib()
    nbytes = _q("salun_conv2d_bf16_wgrad_workspace_bytes", N, H, W, C, K, R, stride, pad)
    if nbytes == 0:
        raise ValueError(f"bf16 backward-weight: unsupported shape C={C} K={K} R={R} stride={stride} pad={pad}")
    ws = workspace(nbytes, x.device)
    dw = out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    if nbias_out is not None and tuple(nbias_out.shape) != (N, K):
        raise ValueError(f"nbias_out must be [{N}, {K}]")
    check(L.salun_conv2d_bf16_backward_weight_ex(_dev(x, torch.bfloat16, "x"), _dev(dy, torch.bfloat16, "dy"),
                                                 _dev(dw, torch.float32, "dw"), _dev(bias_out, torch.float32, "db", True),
                                                 _dev(nbias_out, torch.float32, "dnb", True),
                                                 N, H, W, C, K, R, stride, pad, int(bool(accumulate and out is not None)),
                                                 c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()),
          "salun_conv2d_bf16_backward_weight_ex")
    return dw


def colsum_bf16(dy: torch.Tensor, images: int, bias_out: Optional[torch.Tensor] = None,
                nbias_out: Optional[torch.Tensor] = None, accumulate: bool = False) -> None:
    """dy bf16 [..., K] contiguous (rows = images x pixels): bias_out[K] (=, or += with accumulate) and / or
    nbias_out[images, K] (overwritten) receive its column sums / per-image column sums (fp32)."""
    K = dy.shape[-1]
    M = dy.numel() // K
    ws = workspace(_q("salun_colsum_bf16_workspace_bytes", K), dy.device)
    check(_lib.lib().salun_colsum_bf16(_dev(dy, torch.bfloat16, "dy"), _dev(bias_out, torch.float32, "db", True),
                                       _dev(nbias_out, torch.float32, "dnb", True), c_int64(M), K, int(images),
                                       int(bool(accumulate)), c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()),
          "salun_colsum_bf16")


# ----------------------------------------------------------------------------- K16
def gemm_bf16_supported(M: int, N: int, K: int) -> bool:
    return bool(_q("salun_gemm_bf16_supported", int(M), int(N), int(K)))


def gemm_bf16_nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 addend: Optional[torch.Tensor] = None, variant: int = 0) -> torch.Tensor:
    """y[M, N] = x[M, K] . w[N, K]^T (+ bias fp32 [N]) (+ addend bf16 [M, N]); x, w contiguous bf16 (csrc/salun_gemm.hip)."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().salun_gemm_bf16_nt(_dev(x, torch.bfloat16, "x"), _dev(w, torch.bfloat16, "w"),
                                        _dev(bias, torch.float32, "bias", True), _dev(addend, torch.bfloat16, "addend", True),
                                        c_void_p(y.data_ptr()), c_int64(M), N, K, int(variant), _stream()),
          "salun_gemm_bf16_nt")
    return y


def gemm_bf16_tn_supported(M: int, Na: int, Nb: int) -> bool:
    return bool(_lib.lib().salun_gemm_bf16_tn_supported(c_int64(M), int(Na), int(Nb)))


def gemm_bf16_tn(dy: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False,
                 variant: int = 0) -> torch.Tensor:
    """dw[Na, Nb] fp32 (+= into `out` when accumulate) = dy[M, Na]^T . x[M, Nb]; dy, x contiguous bf16."""
    M, Na = dy.shape
    Nb = x.shape[1]
    assert x.shape[0] == M
    L = _lib.lib()
    if not L.salun_gemm_bf16_tn_supported(c_int64(M), Na, Nb):
        raise ValueError(f"gemm_bf16_tn: unsupported shape M={M} Na={Na} Nb={Nb}")
    nbytes = L.salun_gemm_bf16_tn_workspace_bytes(c_int64(M), Na, Nb, int(variant))
    ws = workspace(nbytes, x.device) if nbytes else None
    dw = out if out is not None else torch.empty((Na, Nb), dtype=torch.float32, device=x.device)
    check(L.salun_gemm_bf16_tn(_dev(dy, torch.bfloat16, "dy"), _dev(x, torch.bfloat16, "x"), _dev(dw, torch.float32, "dw"),
                               c_int64(M), Na, Nb, int(bool(accumulate and out is not None)), int(variant),
                               c_void_p(ws.data_ptr() if ws is not None else None), c_size_t(nbytes), _stream()),
          "salun_gemm_bf16_tn")
    return dw


def pack_bf16(w: torch.Tensor, transposed: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [N, K] master weights -> bf16 [N, K] (or [K, N] with `transposed`): the images gemm_bf16_nt reads."""
    N, K = w.shape
    PACK_CALLS[0] += 1
    if out is None:
        out = torch.empty((K, N) if transposed else (N, K), dtype=torch.bfloat16, device=w.device)
    check(_lib.lib().salun_pack_bf16(_dev(w, torch.float32, "w"), _dev(out, torch.bfloat16, "wp"), N, K, int(bool(transposed)),
                                     _stream()), "salun_pack_bf16")
    return out


# ----------------------------------------------------------------------------- K12
def gn_bf16_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool):
    """x [N, H, W, C] bf16 contiguous -> (y, mr, ab): y = [silu](GroupNorm(x)); mr / ab feed gn_bf16_backward."""
    N, H, W, C = x.shape
    L = _lib.lib()
    ws = workspace(_q("salun_gn_bf16_workspace_bytes", N, C, H * W, groups), x.device)
    y = torch.empty_like(x)
    mr = torch.empty((N, groups, 2), dtype=torch.float32, device=x.device)
    ab = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
    check(L.salun_gn_bf16_forward(_dev(x, torch.bfloat16, "x"), _dev(gamma, torch.float32, "gamma"),
                                  _dev(beta, torch.float32, "beta"), c_void_p(y.data_ptr()), c_void_p(mr.data_ptr()),
                                  c_void_p(ab.data_ptr()), N, C, H * W, groups, c_double(eps), int(bool(silu)),
                                  c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()), "salun_gn_bf16_forward")
    return y, mr, ab


def gn_bf16_backward(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mr: torch.Tensor, ab: torch.Tensor,
                     groups: int, silu: bool, dgamma: torch.Tensor, dbeta: torch.Tensor, accumulate: bool) -> torch.Tensor:
    N, H, W, C = x.shape
    L = _lib.lib()
    ws = workspace(_q("salun_gn_bf16_workspace_bytes", N, C, H * W, groups), x.device)
    dx = torch.empty_like(x)
    check(L.salun_gn_bf16_backward(_dev(dy, torch.bfloat16, "dy"), _dev(x, torch.bfloat16, "x"),
                                   _dev(gamma, torch.float32, "gamma"), _dev(mr, torch.float32, "mr"),
                                   _dev(ab, torch.float32, "ab"), c_void_p(dx.data_ptr()),
                                   _dev(dgamma, torch.float32, "dgamma"), _dev(dbeta, torch.float32, "dbeta"), N, C, H * W,
                                   groups, int(bool(silu)), int(bool(accumulate)), c_void_p(ws.data_ptr()),
                                   c_size_t(ws.numel()), _stream()), "salun_gn_bf16_backward")
    return dx


# ----------------------------------------------------------------------------- K13
def attn_supported(D: int) -> bool:
    return bool(_q("salun_attn_supported", int(D)))


def _tok_view(t: torch.Tensor, name: str):
    """[B, tokens, H, D] bf16 view -> (pointer, batch stride, token stride); head stride D, channel stride 1."""
    if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 4:
        raise TypeError(f"{name}: expected a 4-D bfloat16 device tensor [B, tokens, H, D]")
    B, N, H, D = t.shape
    if t.stride(3) != 1 or (H > 1 and t.stride(2) != D):
        raise ValueError(f"{name}: heads must be adjacent runs of D contiguous channels")
    return c_void_p(t.data_ptr()), ctypes.c_longlong(t.stride(0) if B > 1 else N * t.stride(1)), c_int(t.stride(1))


def attn_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, need_lse: bool = True):
    """softmax(scale * q k^T) v over [B, tokens, H, D] bf16 views -> (o [B, Nq, H, D] contiguous, lse [B*H, Nq] or None)."""
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    o = torch.empty((B, Nq, H, D), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((B * H, Nq), dtype=torch.float32, device=q.device) if need_lse else None
    qp, qb, ql = _tok_view(q, "q")
    kp, kb, kl = _tok_view(k, "k")
    vp, vb, vl = _tok_view(v, "v")
    op, ob, ol = _tok_view(o, "o")
    check(_lib.lib().salun_attn_forward(qp, kp, vp, op, c_void_p(lse.data_ptr() if need_lse else None), B, H, Nq, Nk, D,
                                        qb, ql, kb, kl, vb, vl, ob, ol, c_double(scale), _stream()), "salun_attn_forward")
    return o, lse


def attn_backward(q, k, v, o, d_o, lse, scale: float):
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    dq = torch.empty((B, Nq, H, D), dtype=torch.bfloat16, device=q.device)
    dk = torch.empty((B, Nk, H, D), dtype=torch.bfloat16, device=q.device)
    dv = torch.empty((B, Nk, H, D), dtype=torch.bfloat16, device=q.device)
    dsum = torch.empty((B * H, Nq), dtype=torch.float32, device=q.device)
    qp, qb, ql = _tok_view(q, "q")
    kp, kb, kl = _tok_view(k, "k")
    vp, vb, vl = _tok_view(v, "v")
    op, ob, ol = _tok_view(o, "o")
    dp, db, dl = _tok_view(d_o, "d_o")
    check(_lib.lib().salun_attn_backward(qp, kp, vp, op, dp, _dev(lse, torch.float32, "lse"), c_void_p(dq.data_ptr()),
                                         c_void_p(dk.data_ptr()), c_void_p(dv.data_ptr()), c_void_p(dsum.data_ptr()),
                                         B, H, Nq, Nk, D, qb, ql, kb, kl, vb, vl, ob, ol, db, dl, c_double(scale), _stream()),
          "salun_attn_backward")
    return dq, dk, dv


class _Attn(FastFunction):
    @staticmethod
    def forward(ctx, q, k, v, scale):
        o, lse = attn_forward(q, k, v, scale, need_lse=True)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = float(scale)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        if d_o.stride(3) != 1 or d_o.stride(2) != d_o.shape[3]:
            d_o = d_o.contiguous()
        dq, dk, dv = attn_backward(q, k, v, o, d_o.to(torch.bfloat16), lse, ctx.scale)
        return dq, dk, dv, None


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """Differentiable fused attention over [B, tokens, H, D] bf16 views (csrc/salun_attn.hip, K13)."""
    return _Attn.apply(q, k, v, float(scale))


# ----------------------------------------------------------------------------- K14
class _LayerNorm16(FastFunction):
    """LayerNorm over the last dimension of a bf16 token tensor (csrc/salun_tok_bf16.hip): bf16 in / out, fp32 stats."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[-1]
        xc = x.contiguous()
        rows = xc.numel() // C
        y = torch.empty_like(xc)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        check(_lib.lib().salun_ln_bf16_forward(_dev(xc, torch.bfloat16, "x"), _dev(weight, torch.float32, "gamma"),
                                               _dev(bias, torch.float32, "beta"), c_void_p(y.data_ptr()),
                                               c_void_p(stats.data_ptr()), c_int64(rows), C, c_double(eps), _stream()),
              "salun_ln_bf16_forward")
        ctx.save_for_backward(xc, stats)
        ctx.params = (weight, bias)  # the Parameter objects: gradsink's destinations, and gamma's values in backward (plain
        return y                     # references instead of saved tensors: norm._FusedGN16 says why)

    @staticmethod
    def backward(ctx, dy):
        from . import gradsink
        xc, stats = ctx.saved_tensors
        weight, bias = ctx.params
        C = xc.shape[-1]
        rows = xc.numel() // C
        dyc = dy.to(torch.bfloat16).contiguous()
        L = _lib.lib()
        ws = workspace(_q("salun_ln_bf16_workspace_bytes", rows, C), xc.device)
        gw, gb = gradsink.sink(ctx.params[0]), gradsink.sink(ctx.params[1])
        sunk = gw is not None and gb is not None
        if not sunk:
            gw, gb = torch.empty_like(weight), torch.empty_like(bias)
        dx = torch.empty_like(xc)
        check(L.salun_ln_bf16_backward(_dev(dyc, torch.bfloat16, "dy"), _dev(xc, torch.bfloat16, "x"),
                                       _dev(weight, torch.float32, "gamma"), _dev(stats, torch.float32, "stats"),
                                       c_void_p(dx.data_ptr()), _dev(gw, torch.float32, "dgamma"),
                                       _dev(gb, torch.float32, "dbeta"), c_int64(rows), C, int(sunk),
                                       c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()), "salun_ln_bf16_backward")
        return dx, (None if sunk else gw), (None if sunk else gb), None


def layer_norm_bf16(x: torch.Tensor, ln: "torch.nn.LayerNorm") -> torch.Tensor:
    return _LayerNorm16.apply(x, ln.weight, ln.bias, float(ln.eps))


class _Geglu16(FastFunction):
    """out = h[..., :F] * gelu(h[..., F:]) on a bf16 tensor (csrc/salun_tok_bf16.hip)."""

    @staticmethod
    def forward(ctx, h):
        hc = h.contiguous()
        F2 = hc.shape[-1]
        rows = hc.numel() // F2
        out = torch.empty(hc.shape[:-1] + (F2 // 2,), dtype=torch.bfloat16, device=h.device)
        check(_lib.lib().salun_geglu_bf16_forward(_dev(hc, torch.bfloat16, "h"), c_void_p(out.data_ptr()), c_int64(rows),
                                                  F2 // 2, _stream()), "salun_geglu_bf16_forward")
        ctx.save_for_backward(hc)
        return out

    @staticmethod
    def backward(ctx, dy):
        (hc,) = ctx.saved_tensors
        F2 = hc.shape[-1]
        rows = hc.numel() // F2
        dyc = dy.to(torch.bfloat16).contiguous()
        dh = torch.empty_like(hc)
        check(_lib.lib().salun_geglu_bf16_backward(_dev(hc, torch.bfloat16, "h"), _dev(dyc, torch.bfloat16, "dy"),
                                                   c_void_p(dh.data_ptr()), c_int64(rows), F2 // 2, _stream()),
              "salun_geglu_bf16_backward")
        return dh


def geglu_bf16(h: torch.Tensor) -> torch.Tensor:
    return _Geglu16.apply(h)


# ------------------------------------------------------------------- fused BatchNorm
def bn_forward(x, res, gamma, beta, running_mean, running_var, training, momentum, eps, relu,
               num_batches_tracked=None):
    """-> (y, save_mean, save_invstd) or None when the shape is outside the kernel's domain (HW % 4 != 0)."""
    N, C, H, W = x.shape
    L = _lib.lib()
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(L.salun_bn_workspace_bytes(C), x.device)
    rc = L.salun_bn_forward(_dev(x, torch.float32, "x"), _dev(res, torch.float32, "res", True), c_void_p(y.data_ptr()),
                            _dev(gamma, torch.float32, "weight"), _dev(beta, torch.float32, "bias"),
                            _dev(running_mean, torch.float32, "running_mean", True),
                            _dev(running_var, torch.float32, "running_var", True),
                            _dev(num_batches_tracked, torch.int64, "num_batches_tracked", True),
                            c_void_p(mean.data_ptr()),
                            c_void_p(invstd.data_ptr()), N, C, H * W, int(bool(training)), c_double(momentum),
                            c_double(eps), int(bool(relu)), c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream())
    if rc == _lib.SALUN_EINVAL:
        return None
    check(rc, "salun_bn_forward")
    return y, mean, invstd


def bn_backward(dy, y, x, gamma, mean, invstd, training, relu, want_dres, gamma_grad_acc=None, beta_grad_acc=None):
    """-> (dx, dres or None, dgamma, dbeta); `*_grad_acc` (optional) += dgamma / dbeta in the same launch."""
    N, C, H, W = x.shape
    L = _lib.lib()
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(L.salun_bn_workspace_bytes(C), x.device)
    check(L.salun_bn_backward(_dev(dy, torch.float32, "dy"), _dev(y, torch.float32, "y", True),
                              _dev(x, torch.float32, "x"), _dev(gamma, torch.float32, "weight"),
                              _dev(mean, torch.float32, "mean"), _dev(invstd, torch.float32, "invstd"),
                              c_void_p(dx.data_ptr()), _dev(dres, torch.float32, "dres", True),
                              c_void_p(dgamma.data_ptr()), c_void_p(dbeta.data_ptr()),
                              _dev(gamma_grad_acc, torch.float32, "gamma.grad", True),
                              _dev(beta_grad_acc, torch.float32, "beta.grad", True), N, C, H * W,
                              int(bool(training)), int(bool(relu)), c_void_p(ws.data_ptr()), c_size_t(ws.numel()),
                              _stream()), "salun_bn_backward")
    return dx, dres, dgamma, dbeta


# ------------------------------------------------------------------- fused GroupNorm
def gn_forward(x, gamma, beta, groups, eps, silu):
    """-> (z, save_mean, save_rstd) or None when the shape is outside the kernel's domain."""
    N, C, H, W = x.shape
    z = torch.empty_like(x)
    mean = torch.empty(N * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N * groups, dtype=torch.float32, device=x.device)
    rc = _lib.lib().salun_gn_forward(_dev(x, torch.float32, "x"), c_void_p(z.data_ptr()),
                                     _dev(gamma, torch.float32, "weight"), _dev(beta, torch.float32, "bias"),
                                     c_void_p(mean.data_ptr()), c_void_p(rstd.data_ptr()), N, C, H * W, int(groups),
                                     c_double(eps), int(bool(silu)), _stream())
    if rc == _lib.SALUN_EINVAL:
        return None
    check(rc, "salun_gn_forward")
    return z, mean, rstd


def gn_backward(dz, x, gamma, beta, mean, rstd, groups, silu, gamma_grad_acc=None, beta_grad_acc=None, addend=None,
                nk_sum: bool = False, csum: bool = False, csum_acc=None):
    """-> (dx, dgamma, dbeta), or with `nk_sum` -> (dx, dgamma, dbeta, nk [N, C], csum [C] | None).
    addend [N,C,H,W]: added to dx in the kernel; nk = sum_hw dx per (image, channel); csum = sum_n nk (returned with
    `csum=True`, and / or added into `csum_acc`)."""
    N, C, H, W = x.shape
    L = _lib.lib()
    dx = torch.empty_like(x)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(L.salun_gn_workspace_bytes(N, C), x.device)
    if addend is not None and addend.shape != x.shape:
        raise ValueError("gn_backward: addend must have the shape of x")
    want_nk = bool(nk_sum or csum or csum_acc is not None)
    nk = torch.empty((N, C), dtype=torch.float32, device=x.device) if want_nk else None
    cs = torch.empty(C, dtype=torch.float32, device=x.device) if csum else None
    check(L.salun_gn_backward_fused(_dev(dz, torch.float32, "dz"), _dev(x, torch.float32, "x"),
                                    _dev(gamma, torch.float32, "weight"), _dev(beta, torch.float32, "bias"),
                                    _dev(mean, torch.float32, "mean"), _dev(rstd, torch.float32, "rstd"),
                                    _dev(addend, torch.float32, "addend", True),
                                    c_void_p(dx.data_ptr()), c_void_p(dgamma.data_ptr()), c_void_p(dbeta.data_ptr()),
                                    _dev(gamma_grad_acc, torch.float32, "weight.grad", True),
                                    _dev(beta_grad_acc, torch.float32, "bias.grad", True),
                                    _dev(nk, torch.float32, "nk_sum", True), _dev(cs, torch.float32, "csum", True),
                                    _dev(csum_acc, torch.float32, "csum_acc", True), N, C, H * W, int(groups),
                                    int(bool(silu)), c_void_p(ws.data_ptr()), c_size_t(ws.numel()), _stream()),
          "salun_gn_backward_fused")
    if want_nk:
        return dx, dgamma, dbeta, nk, cs
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------ K9 / K10
def proximal_step(p: torch.Tensor, p0: torch.Tensor, ratio: int, scratch: Optional[torch.Tensor] = None,
                  scratch_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In place soft-threshold of `p` towards `p0` with threshold = the ratio-th smallest |p - p0|
    (RL_pro.py:52-60).  Returns the threshold as a 1-element device tensor (no host sync)."""
    PARAM_EPOCH[0] += 1
    n = p.numel()
    if ratio < 1:
        raise IndexError("index -1 is out of bounds for dimension 0 with size 0")  # reference: topk(.., 0)[0][-1]
    if ratio > n:
        raise RuntimeError("selected index k out of range")  # reference: torch.topk with k > n
    L = _lib.lib()
    d = scratch if scratch is not None else torch.empty(n, dtype=torch.float32, device=p.device)
    check(L.salun_param_diff(_dev(p, torch.float32, "p"), _dev(p0, torch.float32, "p0"), _dev(d, torch.float32, "d"),
                             c_int64(n), _stream()), "salun_param_diff")
    # k-th largest |d| with k = n - ratio + 1 == ratio-th smallest; only the threshold is needed: no mask is written
    mask_topk(d, [n - ratio + 1], flags=_lib.SALUN_TOPK_VALUES_ONLY)
    tau = mask_topk_thresholds(p.device, 1)
    check(L.salun_soft_threshold_step(_dev(p, torch.float32, "p"), _dev(p0, torch.float32, "p0"),
                                      c_void_p(tau.data_ptr()), c_int64(n), _stream()), "salun_soft_threshold_step")
    return tau


def ewc_penalty_grad(p: torch.Tensor, p_star: torch.Tensor, F: torch.Tensor, g: torch.Tensor, lam: float
                     ) -> torch.Tensor:
    """g += grad of lam * sum F (p - p*)^2; returns device tensor [lam * S, S]."""
    n = p.numel()
    L = _lib.lib()
    out = torch.empty(2, dtype=torch.float32, device=p.device)
    ws = workspace(L.salun_ewc_workspace_bytes(c_int64(n)), p.device)
    check(L.salun_ewc_penalty_grad(_dev(p, torch.float32, "p"), _dev(p_star, torch.float32, "p_star"),
                                   _dev(F, torch.float32, "F"), _dev(g, torch.float32, "g"), c_double(lam),
                                   c_void_p(out.data_ptr()), c_int64(n), c_void_p(ws.data_ptr()),
                                   c_size_t(ws.numel()), _stream()), "salun_ewc_penalty_grad")
    return out


# ----------------------------------------------------------------------------- K0
def image_batch(data: torch.Tensor, idx: torch.Tensor, crop: Optional[torch.Tensor] = None,
                flip: Optional[torch.Tensor] = None, pad: int = 4, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(num,H,W,C) u8 resident dataset -> (B,C,H,W) fp32 batch with crop/flip/ToTensor fused."""
    num, H, W, C = data.shape
    B = idx.numel()
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=data.device)
    check(_lib.lib().salun_image_batch(_dev(data, torch.uint8, "data"), _dev(idx, torch.int64, "idx"),
                                       _dev(crop, torch.int32, "crop", True), _dev(flip, torch.uint8, "flip", True),
                                       c_void_p(out.data_ptr()), c_int64(B), c_int(H), c_int(W), c_int(C), c_int(pad),
                                       _stream()), "salun_image_batch")
    return out


# --------------------------------------------------------------------- generators
def fill_uniform(n: int, seed: int, lo: float = 0.0, hi: float = 1.0, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(_lib.lib().salun_fill_uniform(c_void_p(out.data_ptr()), c_int64(n), c_uint64(seed), c_double(lo),
                                        c_double(hi), _stream()), "salun_fill_uniform")
    return out


def fill_normal(n: int, seed: int, mean: float = 0.0, std: float = 1.0, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(_lib.lib().salun_fill_normal(c_void_p(out.data_ptr()), c_int64(n), c_uint64(seed), c_double(mean),
                                       c_double(std), _stream()), "salun_fill_normal")
    return out


def fill_u8(n: int, seed: int, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.uint8, device=device)
    check(_lib.lib().salun_fill_u8(c_void_p(out.data_ptr()), c_int64(n), c_uint64(seed), _stream()), "salun_fill_u8")
    return out
