"""A ResNet BasicBlock as ONE autograd node on the HIP kernels.

    out = relu( bn2(conv2( relu(bn1(conv1(x))) )) + skip ),   skip = x  |  bn_d(conv_d(x))
    (reference block: Classification/models/ResNet.py:93-125)

Forward is 2-3 MFMA convolutions + 2-3 fused BN launches; backward walks the same chain by hand:
the residual branch's gradient is folded into conv1's backward-data epilogue (`addend`) instead of a separate
add pass over the activation, and all weight / BN gradients are accumulated by their kernels directly into the
parameters' `.grad` storage (gradsink.py) — no AccumulateGrad launches.  Per block that removes one
3-pass elementwise add over the block input and 6-9 small adds, plus the autograd bookkeeping of ~10 nodes.
In forward the projection branch bn_d(conv_d(x)) of a strided block (a 1x1 stride-2 convolution at 13 - 50 TFLOP/s and a
small BatchNorm: launches that leave most of the chip idle) is issued on the side stream beside conv1 -> bn1 -> conv2
(round 6: +1.2 % on the ResNet-18 step; bit-identical results, tests/test_classification_gpu.py).

`fused_basic_block(blk, x)` returns None when the block cannot take this path (non-fp32 / CPU tensors, autocast,
SyncBatchNorm, shapes outside the convolution kernels' tiling domain — probed once per input shape); the module
then runs its ordinary forward.
"""
from __future__ import annotations

import torch

from .fastfn import FastFunction
import torch.nn as nn

import os

from . import dist as sdist
from . import draws, gradsink, ops, streams

# Backward-weight and backward-data of one convolution depend on the same dY and on nothing of each other.  The
# backward-weight kernel runs ONE wave per SIMD (register budget) and leaves LDS for a second workgroup, so issuing it
# on a side stream lets the two kernels share the CUs — the matrix pipes idle less than when either runs alone.
# SALUN_WGRAD_OVERLAP=0 keeps everything on one stream.
OVERLAP_WGRAD = os.environ.get("SALUN_WGRAD_OVERLAP", "1") != "0"
FWD_SHORTCUT_BESIDE = os.environ.get("SALUN_FWD_SHORTCUT_BESIDE", "1") != "0"  # _BasicBlockFn.forward


class overlap_disabled:
    """Context manager: keep backward-weight on the main stream.  Needed whenever something else than the
    convolution kernels writes a parameter's `.grad` during the same backward pass — e.g. the l1 penalty of FT_l1 /
    GA_l1 (`_steps.l1_regularization`), whose AccumulateGrad `w.grad.add_()` runs on the main stream and would race
    with a side-stream `salun_conv2d_backward_weight` accumulating into the same slice."""

    def __enter__(self):
        global OVERLAP_WGRAD
        self._prev = OVERLAP_WGRAD
        OVERLAP_WGRAD = False
        return self

    def __exit__(self, *exc):
        global OVERLAP_WGRAD
        OVERLAP_WGRAD = self._prev
        return False


def reset_join_state() -> None:
    """Forget a pending end-of-backward join (only needed after a backward pass was aborted by an exception) and make
    the current streams wait for whatever the side streams still have in flight."""
    _join_queued.clear()
    _held.clear()
    for dev, side in _side_streams.items():
        torch.cuda.current_stream(dev).wait_stream(side)
_side_streams: dict = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = streams.concurrent_stream(device)
    return s


class _on_side:
    """`with torch.cuda.stream(side)` for a call that allocates nothing with torch (the gradient goes into `.grad` storage):
    the wrappers of ops.py are handed the raw handle instead (ops._STREAM_OVERRIDE) — ~10 us of host time less per use, 20
    uses per ResNet-18 step, which the data-parallel step (hooks and slice launches on top) feels.  With `alloc` (no sink:
    the result is a fresh tensor that must belong to the side stream) the ordinary stream context is entered."""

    __slots__ = ("side", "ctx")

    def __init__(self, side, alloc: bool):
        self.side = side
        self.ctx = torch.cuda.stream(side) if alloc else None

    def __enter__(self):
        if self.ctx is not None:
            return self.ctx.__enter__()
        ops._STREAM_OVERRIDE[0] = self.side.cuda_stream

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        ops._STREAM_OVERRIDE[0] = None
        return False


_join_queued: set = set()
# Gradient tensors a side-stream kernel is still reading.  autograd OWNS a gradient buffer once every node it was handed
# to has returned, and accumulates further contributions into it IN PLACE when nobody else holds it
# (InputBuffer::add: `old.add_(new)` if use_count == 1) — on the main stream, while the side stream may still be reading
# it: `record_stream` guards against reuse after free, not against that write.  A held reference makes the engine
# accumulate out of place.  Found in round 4 (the first AttnBlock's proj_out weight gradient changed from run to run
# once the attention's backward became short enough for the residual's accumulation to overtake the 1x1 backward-weight).
_held: dict = {}


def hold_until_join(t: torch.Tensor) -> None:
    """Keep `t` referenced until the side stream has passed the kernels enqueued on it so far, at the latest until the
    end-of-backward join.  One event per EIGHT tensors (an event per tensor was 11 us of host time on each of ~470
    backward-weight launches of an SD step): a batch is released when the event recorded behind its last member has
    completed.  Inside a stream capture events cannot be queried: the references simply live until the join."""
    st = _held.get(t.device)
    if st is None:
        st = _held[t.device] = {"open": [], "closed": []}
    st["open"].append(t)
    if len(st["open"]) < 8 or torch.cuda.is_current_stream_capturing():
        return
    ev = torch.cuda.Event()
    ev.record(_side_stream(t.device))
    st["closed"].append((ev, st["open"]))
    st["open"] = []
    closed = st["closed"]
    while closed and closed[0][0].query():
        closed.pop(0)


def release_held(device) -> None:
    """After the main stream has been made to wait for the side stream: nothing is being read there any more."""
    _held.pop(device, None)


def _join_at_end_of_backward(device: torch.device) -> None:
    """Single process: the main stream waits for the side stream ONCE, when the whole backward pass has been issued
    (autograd's end-of-backward callback) — so after `loss.backward()` returns, gradients are ordered on the current
    stream as usual, and inside the pass the weight-gradient kernels of one block overlap the next block's work.
    Caveat: if a backward pass dies with an exception the engine drops its callbacks; call `reset_join_state()` (or
    set SALUN_WGRAD_OVERLAP=0) before reusing the process after such a failure."""
    if device in _join_queued:
        return
    _join_queued.add(device)

    def _join():
        _join_queued.discard(device)
        torch.cuda.current_stream(device).wait_stream(_side_stream(device))
        _held.pop(device, None)

    torch.autograd.Variable._execution_engine.queue_callback(_join)


def _bn_args(bn: nn.BatchNorm2d):
    return bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps


class _BasicBlockFn(FastFunction):
    @staticmethod
    def forward(ctx, x, blk, w1, g1, b1, w2, g2, b2, wd, gd, bd):
        s = blk.conv1.stride[0]
        N, C, H, W = x.shape
        P, Q = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        training = blk.bn1.training
        cd = md = idd = None
        fork = None
        if wd is not None:
            def shortcut():
                cd_ = ops.conv2d_forward(x, wd, None, s, 0, P, Q)
                return (cd_,) + tuple(ops.bn_forward(cd_, None, gd, bd, *_bn_args(blk.downsample[1]), False,
                                                     blk.downsample[1].num_batches_tracked))
            if FWD_SHORTCUT_BESIDE and OVERLAP_WGRAD and x.is_cuda and not torch.cuda.is_current_stream_capturing():
                # the 1x1 stride-2 projection + its BatchNorm depend on x alone: issued on the (idle, in forward)
                # backward-weight side stream beside conv1 -> bn1 -> conv2, joined before bn2 adds the result
                main, fork = torch.cuda.current_stream(x.device), _side_stream(x.device)
                fork.wait_stream(main)
                with torch.cuda.stream(fork):
                    cd, skip, md, idd = shortcut()
                x.record_stream(fork)
        c1 = ops.conv2d_forward(x, w1, None, s, 1, P, Q)
        y1, m1, i1 = ops.bn_forward(c1, None, g1, b1, *_bn_args(blk.bn1), True, blk.bn1.num_batches_tracked)
        c2 = ops.conv2d_forward(y1, w2, None, 1, 1, P, Q)
        if wd is None:
            skip = x
        elif fork is None:
            cd, skip, md, idd = shortcut()
        else:
            main.wait_stream(fork)
            for t in (cd, skip, md, idd):  # allocated under the side stream, used (and freed) on this one from here on
                if t is not None:
                    t.record_stream(main)
        out, m2, i2 = ops.bn_forward(c2, skip, g2, b2, *_bn_args(blk.bn2), True, blk.bn2.num_batches_tracked)
        ctx.save_for_backward(x, c1, y1, c2, out, cd, w1, g1, b1, w2, g2, b2, wd, gd, bd, m1, i1, m2, i2, md, idd)
        ctx.cfg = (s, training)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, c1, y1, c2, out, cd, w1, g1, b1, w2, g2, b2, wd, gd, bd, m1, i1, m2, i2, md, idd = ctx.saved_tensors
        s, training = ctx.cfg
        dout = dout.contiguous()
        grads = {}

        def bn_bwd(dy, y, xin, g, b, m, i, relu, want_dres):
            gw, gb = gradsink.sink(g), gradsink.sink(b)
            if gw is None or gb is None:
                gw = gb = None
            dx, dres, dg, db = ops.bn_backward(dy, y, xin, g, m, i, training, relu, want_dres, gw, gb)
            if gw is not None:
                gradsink.arrived(g)
                gradsink.arrived(b)
                dg = db = None
            return dx, dres, dg, db

        main = torch.cuda.current_stream(dout.device)
        side = _side_stream(dout.device) if OVERLAP_WGRAD else None

        def wgrad(xin, dy, w, stride, pad):
            dst = gradsink.sink(w)
            if side is None:
                dw = ops.conv2d_backward_weight(xin, dy, w.shape, stride, pad, out=dst, accumulate=True)
            else:
                side.wait_stream(main)  # dy (and everything before it) is complete for the side stream
                with _on_side(side, dst is None):
                    dw = ops.conv2d_backward_weight(xin, dy, w.shape, stride, pad, out=dst, accumulate=True, shared=True)
                for t in (xin, dy):  # freed when this backward returns: keep the memory until the side stream is done
                    t.record_stream(side)
                hold_until_join(dy)  # ... and keep autograd from accumulating into it in place meanwhile
                if dst is None and dw is not None:
                    dw.record_stream(main)
            if dst is not None:
                gradsink.arrived(w)
                return None
            return dw

        # parameters are visited last-to-first so gradient-arrival hooks see the order autograd would produce
        dc2, dskip, dg2, db2 = bn_bwd(dout, out, c2, g2, b2, m2, i2, True, True)
        dwd = dgd = dbd = None
        dxd = dskip
        if wd is not None:
            # (this branch on a THIRD stream beside dgrad(conv2) -> bn1 backward, the mirror of what forward does, was
            # measured and dropped: 117.6 -> 115.8 steps/s, and 8.67 -> 9.77 ms under data parallel — backward already
            # shares every SIMD between the compute stream and the backward-weight stream)
            dcd, _, dgd, dbd = bn_bwd(dskip, None, cd, gd, bd, md, idd, False, False)
            dwd = wgrad(x, dcd, wd, s, 0)
            dxd = ops.conv2d_backward_data(dcd, wd, x.shape, s, 0) if ctx.needs_input_grad[0] else None
        dw2 = wgrad(y1, dc2, w2, 1, 1)
        dy1 = ops.conv2d_backward_data(dc2, w2, y1.shape, 1, 1)
        dc1, _, dg1, db1 = bn_bwd(dy1, y1, c1, g1, b1, m1, i1, True, False)
        dw1 = wgrad(x, dc1, w1, s, 1)
        dx = ops.conv2d_backward_data(dc1, w1, x.shape, s, 1, addend=dxd) if ctx.needs_input_grad[0] else None
        if side is not None:
            if not all(g is None for g in (dw1, dw2, dwd)):
                # autograd route: AccumulateGrad consumes the returned tensors on the main stream -> join now.
                # (Data parallel needs no join here any more: a gradient slice's all-reduce waits for the side stream
                # itself — dist.BucketedGradReducer._launch.)
                main.wait_stream(side)
                release_held(dout.device)
            else:
                _join_at_end_of_backward(dout.device)
        return dx, None, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd


def _probe(blk, shape, device) -> bool:
    """Can every convolution launch of this block run on the MFMA kernels for this input shape?  One-time dry run
    on uninitialised buffers (only the return codes matter)."""
    N, C, H, W = shape
    s = blk.conv1.stride[0]
    P, Q = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    K = blk.conv1.out_channels
    if (P * Q) % 4 or (H * W) % 4:
        return False
    e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=device)
    x, y1, dy = e(N, C, H, W), e(N, K, P, Q), e(N, K, P, Q)
    convs = [(x, blk.conv1.weight, s, 1), (y1, blk.conv2.weight, 1, 1)]
    if blk.downsample is not None:
        convs.append((x, blk.downsample[0].weight, s, 0))
    for xin, w, st, pad in convs:
        if ops.conv2d_forward(xin, w, None, st, pad, P, Q) is None:
            return False
        if ops.conv2d_backward_data(dy, w, xin.shape, st, pad, addend=xin) is None:
            return False
        if ops.conv2d_backward_weight(xin, dy, w.shape, st, pad) is None:
            return False
    return True


def _structure_ok(blk) -> bool:
    convs = [blk.conv1, blk.conv2] + ([blk.downsample[0]] if blk.downsample is not None else [])
    bns = [blk.bn1, blk.bn2] + ([blk.downsample[1]] if blk.downsample is not None else [])
    if any(type(b) is not nn.BatchNorm2d or not b.affine or not b.track_running_stats or b.momentum is None
           for b in bns):
        return False
    if any(c.bias is not None or c.groups != 1 or c.dilation != (1, 1) or c.weight.dtype != torch.float32
           for c in convs):
        return False
    if len({b.training for b in bns}) != 1:
        return False
    c1, c2 = blk.conv1, blk.conv2
    ok = (c1.kernel_size == (3, 3) and c1.padding == (1, 1) and c1.stride[0] == c1.stride[1] and c1.stride[0] in (1, 2)
          and c2.kernel_size == (3, 3) and c2.padding == (1, 1) and c2.stride == (1, 1))
    if blk.downsample is not None:
        d = blk.downsample[0]
        ok = ok and d.kernel_size == (1, 1) and d.padding == (0, 0) and d.stride == c1.stride
    else:
        ok = ok and c1.stride == (1, 1) and c1.in_channels == c1.out_channels
    return ok


def fused_basic_block(blk, x: torch.Tensor):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4) or torch.is_autocast_enabled():
        return None
    cache = blk.__dict__.setdefault("_fused_shapes", {})
    key = (tuple(x.shape), x.device)
    ok = cache.get(key)
    if ok is None:
        with torch.no_grad():
            ok = cache[key] = _structure_ok(blk) and _probe(blk, tuple(x.shape), x.device)
    if not ok or not _structure_ok(blk):
        return None
    x = x.contiguous()
    d = blk.downsample
    return _BasicBlockFn.apply(x, blk, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight,
                               blk.bn2.weight, blk.bn2.bias, d[0].weight if d is not None else None,
                               d[1].weight if d is not None else None, d[1].bias if d is not None else None)


# =====================================================================================================================
# The diffusion U-Net's ResnetBlock as ONE autograd node (reference block: DDPM/models/diffusion.py:85-128)
#
#     h   = conv1(swish(norm1(x))) + proj[:, :, None, None]        proj = Linear(swish([temb | cemb]))  (outside)
#     out = skip(x) + conv2(dropout(swish(norm2(h))))              skip = identity | 1x1 (nin) | 3x3 convolution
#
# Forward: 2 fused GroupNorm+SiLU launches and 2-3 MFMA convolutions whose epilogues carry the two adds (`nbias` = the
# embedding projection, `addend` = the skip branch) — no element-wise pass over an activation besides dropout.
# Backward: norm2's backward kernel also emits sum_hw of its dx per (image, channel) — that IS the projection's
# gradient, and folded over the batch conv1's bias gradient — and norm1's backward adds the skip branch's gradient
# before its store; conv2's and the skip convolution's bias gradients share one streaming channel sum of dout; weight
# gradients go to the side stream and straight into `.grad` (gradsink), as for the BasicBlock above.
class _DiffusionResnetBlockFn(FastFunction):
    @staticmethod
    def forward(ctx, x, proj, blk, n1w, n1b, w1, b1, n2w, n2b, w2, b2, ws, bs):
        N, C, H, W = x.shape
        G1, G2 = blk.norm1.num_groups, blk.norm2.num_groups
        a1, m1, r1 = ops.gn_forward(x, n1w, n1b, G1, blk.norm1.eps, True)
        c1 = ops.conv2d_forward(a1, w1, b1, 1, 1, H, W, nbias=proj)
        a2, m2, r2 = ops.gn_forward(c1, n2w, n2b, G2, blk.norm2.eps, True)
        p = float(blk.dropout.p) if blk.dropout.training else 0.0
        dkey = None
        if p > 0.0:
            # counter-based keep decisions keyed by the global sample index (draws.py): nothing to save but the key
            dkey = draws.dropout_key()
            a2 = ops.dropout(a2, p, dkey[0], dkey[1], out=a2)
        if ws is not None:
            # (on the side stream beside the norm1 -> conv1 -> norm2 chain, as _BasicBlockFn.forward does with its 1x1
            # stride-2 projection: measured and dropped, 104.4 -> 107.9 ms per DDPM step over three alternated pairs —
            # these convolutions fill the chip, the chain only loses what the skip branch takes)
            sc = ops.conv2d_forward(x, ws, bs, 1, (ws.shape[2] - 1) // 2, H, W)
        else:
            sc = x
        out = ops.conv2d_forward(a2, w2, b2, 1, 1, H, W, addend=sc)
        ctx.save_for_backward(x, a1, c1, a2, n1w, n1b, w1, b1, n2w, n2b, w2, b2, ws, bs, m1, r1, m2, r2)
        ctx.cfg = (G1, G2, p, dkey)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, a1, c1, a2, n1w, n1b, w1, b1, n2w, n2b, w2, b2, ws, bs, m1, r1, m2, r2 = ctx.saved_tensors
        G1, G2, p, dkey = ctx.cfg
        dout = dout.contiguous()
        main = torch.cuda.current_stream(dout.device)
        side = _side_stream(dout.device) if OVERLAP_WGRAD else None
        returned = []  # weight gradients handed back to autograd (no sink): they need the main stream to have joined

        def wgrad(xin, dy, w, pad):
            dst = gradsink.sink(w)
            if side is None:
                dw = ops.conv2d_backward_weight(xin, dy, w.shape, 1, pad, out=dst, accumulate=True)
            else:
                side.wait_stream(main)
                with _on_side(side, dst is None):
                    dw = ops.conv2d_backward_weight(xin, dy, w.shape, 1, pad, out=dst, accumulate=True, shared=True)
                for t in (xin, dy):
                    t.record_stream(side)
                hold_until_join(dy)
                if dst is None and dw is not None:
                    dw.record_stream(main)
            if dst is not None:
                gradsink.arrived(w)
                return None
            returned.append(dw)
            return dw

        def gn_sinks(g, b):
            gw, gb = gradsink.sink(g), gradsink.sink(b)
            return (gw, gb) if gw is not None and gb is not None else (None, None)

        # ---- bias gradients of conv2 and of the skip convolution: one channel sum of dout
        db2 = dbs = None
        s2, ss = gradsink.sink(b2), (gradsink.sink(bs) if bs is not None else None)
        if bs is None and s2 is not None:
            ops.channel_sum(dout, out=s2, accumulate=True)
        else:
            csum = ops.channel_sum(dout)
            if s2 is not None:
                s2.add_(csum)
            else:
                db2 = csum
            if bs is not None:
                if ss is not None:
                    ss.add_(csum)
                else:
                    # never hand ONE tensor to autograd as the gradient of two parameters: AccumulateGrad may adopt it
                    # as `.grad` without a copy, and the two `.grad`s would alias
                    dbs = csum.clone() if db2 is csum else csum
        # ---- conv2
        dw2 = wgrad(a2, dout, w2, 1)
        da2 = ops.conv2d_backward_data(dout, w2, a2.shape, 1, 1)
        if dkey is not None:
            da2 = ops.dropout(da2, p, dkey[0], dkey[1], out=da2)
        # ---- norm2 (+ SiLU): dx = dc1; its per-(image, channel) sums are dproj, their batch sum conv1's bias gradient
        gw, gb = gn_sinks(n2w, n2b)
        s1 = gradsink.sink(b1)
        dc1, dg2, dbt2, nk, cs = ops.gn_backward(da2, c1, n2w, n2b, m2, r2, G2, True, gw, gb, nk_sum=True,
                                                 csum=s1 is None, csum_acc=s1)
        if gw is not None:
            dg2 = dbt2 = None
        db1 = cs  # None when it was added into b1.grad by the kernel
        dproj = nk if ctx.needs_input_grad[1] else None
        # ---- conv1
        dw1 = wgrad(a1, dc1, w1, 1)
        da1 = ops.conv2d_backward_data(dc1, w1, a1.shape, 1, 1)
        # ---- norm1 (+ SiLU) and the skip branch
        gw, gb = gn_sinks(n1w, n1b)
        dws = None
        if ws is None:
            dx, dg1, dbt1 = ops.gn_backward(da1, x, n1w, n1b, m1, r1, G1, True, gw, gb, addend=dout)
        else:
            pad = (ws.shape[2] - 1) // 2
            dws = wgrad(x, dout, ws, pad)
            dxg, dg1, dbt1 = ops.gn_backward(da1, x, n1w, n1b, m1, r1, G1, True, gw, gb)
            dx = ops.conv2d_backward_data(dout, ws, x.shape, 1, pad, addend=dxg)
        if gw is not None:
            dg1 = dbt1 = None
        if side is not None:
            if returned:
                main.wait_stream(side)
                release_held(dout.device)
            else:
                _join_at_end_of_backward(dout.device)
        if not ctx.needs_input_grad[0]:
            dx = None
        return dx, dproj, None, dg1, dbt1, dw1, db1, dg2, dbt2, dw2, db2, dws, dbs


def _diffusion_block_structure_ok(blk) -> bool:
    convs = [blk.conv1, blk.conv2]
    skip = getattr(blk, "conv_shortcut", None) if blk.use_conv_shortcut else getattr(blk, "nin_shortcut", None)
    if blk.in_channels != blk.out_channels:
        if skip is None:
            return False
        convs.append(skip)
    if any(type(n) is not nn.GroupNorm or not n.affine or n.weight.dtype != torch.float32
           for n in (blk.norm1, blk.norm2)):
        return False
    for c in convs:
        if (c.bias is None or c.groups != 1 or c.dilation != (1, 1) or c.stride != (1, 1)
                or c.padding_mode != "zeros" or c.weight.dtype != torch.float32):
            return False
    ok = all(c.kernel_size == (3, 3) and c.padding == (1, 1) for c in (blk.conv1, blk.conv2))
    if len(convs) == 3:
        k = convs[2].kernel_size
        ok = ok and k in ((1, 1), (3, 3)) and convs[2].padding == ((k[0] - 1) // 2,) * 2
    return ok and type(blk.dropout) is draws.CounterDropout and 0.0 <= blk.dropout.p < 1.0 and not blk.dropout.inplace


def _diffusion_block_probe(blk, shape, device) -> bool:
    """Dry run of every launch of the node on uninitialised buffers (only the return codes matter)."""
    N, C, H, W = shape
    K = blk.out_channels
    hw = H * W
    if hw < 4 or hw & (hw - 1) or C // blk.norm1.num_groups > 256 or K // blk.norm2.num_groups > 256:
        return False
    e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=device)
    x, h, proj = e(N, C, H, W), e(N, K, H, W), e(N, K)
    try:
        for t, n in ((x, blk.norm1), (h, blk.norm2)):
            z, m, r = ops.gn_forward(t, n.weight, n.bias, n.num_groups, n.eps, True)
            ops.gn_backward(z, t, n.weight, n.bias, m, r, n.num_groups, True, addend=t, nk_sum=True)
    except Exception:
        return False
    convs = [(x, blk.conv1), (h, blk.conv2)]
    if C != K:
        convs.append((x, blk.conv_shortcut if blk.use_conv_shortcut else blk.nin_shortcut))
    for xin, c in convs:
        pad = c.padding[0]
        if ops.conv2d_forward(xin, c.weight, c.bias, 1, pad, H, W, nbias=proj, addend=h) is None:
            return False
        if ops.conv2d_backward_data(h, c.weight, xin.shape, 1, pad, addend=xin) is None:
            return False
        if ops.conv2d_backward_weight(xin, h, c.weight.shape, 1, pad) is None:
            return False
    return True


def fused_diffusion_resnet_block(blk, x: torch.Tensor, emb_act: torch.Tensor, proj: torch.Tensor = None):
    """`blk(x, emb_act)` as one autograd node, or None when the block cannot take this path (the module then runs
    its ordinary forward)."""
    from . import norm
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4) or torch.is_autocast_enabled() or not norm._FUSED_GN:
        return None
    cache = blk.__dict__.setdefault("_fused_shapes", {})
    key = (tuple(x.shape), x.device)
    ok = cache.get(key)
    if ok is None:
        with torch.no_grad():
            ok = cache[key] = _diffusion_block_structure_ok(blk) and _diffusion_block_probe(blk, tuple(x.shape), x.device)
    if not ok or not _diffusion_block_structure_ok(blk):
        return None
    if proj is None:
        proj = blk.temb_cemb_proj(emb_act)
    if proj.dtype != torch.float32 or tuple(proj.shape) != (x.shape[0], blk.out_channels):
        return None
    skip = None
    if blk.in_channels != blk.out_channels:
        skip = blk.conv_shortcut if blk.use_conv_shortcut else blk.nin_shortcut
    return _DiffusionResnetBlockFn.apply(x.contiguous(), proj.contiguous(), blk, blk.norm1.weight, blk.norm1.bias,
                                         blk.conv1.weight, blk.conv1.bias, blk.norm2.weight, blk.norm2.bias,
                                         blk.conv2.weight, blk.conv2.bias,
                                         skip.weight if skip is not None else None,
                                         skip.bias if skip is not None else None)
