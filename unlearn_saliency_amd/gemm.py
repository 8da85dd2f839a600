"""Host side of K15 (csrc/salun_gemm.hip): the fp32 Linear layers and the fp32 attention of the diffusion U-Nets on this
package's own MFMA GEMM, instead of the library GEMMs behind `nn.Linear` / `scaled_dot_product_attention`.

  linear(x, weight, bias)                     y = x W^T + b        reference: DDPM/models/diffusion.py:85-145,
                                                                   SD/ldm/modules/attention.py:37-75,149-200 (fp32)
  grouped_linear(x, [(weight, bias), ...])    the DDPM's per-ResnetBlock embedding projections (all 22 read the same
                                              [batch, 1024] activation): ONE launch forward, one for the input
                                              gradient (22 chained segments), one for the 22 weight gradients
  attention_f32(q, k, v, scale)               softmax(scale q k^T) v over [B, H, T, D] views of ANY strides — the DDPM's
                                              channel-major AttnBlock tensors (DDPM/models/diffusion.py:148-192) and the
                                              SD [b, n, h, d] projections (attention.py:168-192) are read and written in
                                              place, no transposing copies: GEMM -> row softmax -> GEMM, P kept for backward
  use_salun_linears(model)                    re-classes eligible nn.Linear modules in place (parameters stay views of
                                              the flat arena; state_dict unchanged)

Weight / bias gradients are accumulated by the kernels straight into the parameters' `.grad` slices of the flat arena
(gradsink.py) when that is provably equivalent, else returned to autograd.  fp32 device tensors only; anything else
(CPU tensors of the tests' reference runs, autocast regions) takes the library path of the module it replaces.
"""
from __future__ import annotations

import os

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from .fastfn import FastFunction
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, gradsink, ops
from ._lib import GemmJob, GemmSeg, c_double, c_int, c_int64, c_size_t, c_void_p, check


class Seg:
    """A(i, k) = a[i * a_i + k * a_k], B(j, k) = b[j * b_j + k * b_k] over k in [0, K)."""
    __slots__ = ("a", "b", "K", "a_i", "a_k", "b_j", "b_k")

    def __init__(self, a: int, a_i: int, a_k: int, b: int, b_j: int, b_k: int, K: int):
        self.a, self.a_i, self.a_k, self.b, self.b_j, self.b_k, self.K = a, a_i, a_k, b, b_j, b_k, K


class Job:
    __slots__ = ("c", "M", "N", "ldc", "segs", "bias", "accumulate")

    def __init__(self, c: int, M: int, N: int, ldc: int, segs: Sequence[Seg], bias: int = 0, accumulate: bool = False):
        self.c, self.M, self.N, self.ldc, self.segs, self.bias, self.accumulate = c, M, N, ldc, list(segs), bias, accumulate


def launch(jobs: Sequence[Job], device: torch.device, alpha: float = 1.0, batch: Tuple[int, int] = (1, 1),
           batch_strides: Optional[Sequence[int]] = None) -> None:
    """One `salun_gemm_f32` call.  Pointers are raw device addresses (`tensor.data_ptr()`); the caller keeps the
    tensors alive (they are stream-ordered allocations of the caching allocator)."""
    L = _lib.lib()
    nseg = sum(len(j.segs) for j in jobs)
    if not 1 <= len(jobs) <= _lib.SALUN_GEMM_MAX_JOBS or not 1 <= nseg <= _lib.SALUN_GEMM_MAX_SEGS:
        raise ValueError(f"salun_gemm_f32: {len(jobs)} jobs / {nseg} segments (limits {_lib.SALUN_GEMM_MAX_JOBS} / "
                         f"{_lib.SALUN_GEMM_MAX_SEGS})")
    J = (GemmJob * len(jobs))()
    S = (GemmSeg * nseg)()
    si = 0
    for ji, j in enumerate(jobs):
        J[ji].C, J[ji].bias, J[ji].M, J[ji].N, J[ji].ldc = j.c, (j.bias or None), j.M, j.N, j.ldc
        J[ji].seg0, J[ji].nseg, J[ji].accumulate = si, len(j.segs), int(bool(j.accumulate))
        for s in j.segs:
            S[si].A, S[si].B, S[si].K, S[si].a_i, S[si].a_k, S[si].b_j, S[si].b_k = s.a, s.b, s.K, s.a_i, s.a_k, s.b_j, s.b_k
            si += 1
    bs = (c_int64 * 6)(*[int(v) for v in batch_strides]) if batch_strides is not None else None
    nbytes = L.salun_gemm_f32_workspace_bytes(J, len(jobs), S, nseg, batch[0], batch[1])
    ws = ops.workspace(nbytes, device, "gemm") if nbytes else None
    check(L.salun_gemm_f32(J, len(jobs), S, nseg, batch[0], batch[1], bs, c_double(alpha),
                           c_void_p(ws.data_ptr() if ws is not None else None), c_size_t(nbytes),
                           c_void_p(torch.cuda.current_stream().cuda_stream)), "salun_gemm_f32")


def _f32_dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32:
        raise TypeError(f"{name}: expected an fp32 device tensor, got {t.dtype} on {t.device} (no CPU fallback)")
    return t


def mm_nt(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
          alpha: float = 1.0, accumulate: bool = False) -> torch.Tensor:
    """out[M, N] (+)= alpha * a[M, K] . b[N, K]^T (+ bias[N]) for 2-D fp32 views of any strides; `out` must have unit
    column stride."""
    _f32_dev(a, "a"), _f32_dev(b, "b")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K, (a.shape, b.shape)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    launch([Job(out.data_ptr(), M, N, out.stride(0),
                [Seg(a.data_ptr(), a.stride(0), a.stride(1), b.data_ptr(), b.stride(0), b.stride(1), K)],
                bias.data_ptr() if bias is not None else 0, accumulate)], a.device, alpha)
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[N] (+)= sum_m x[m, n] for a row-major 2-D view (unit column stride)."""
    _f32_dev(x, "x")
    M, N = x.shape
    assert x.stride(1) == 1
    L = _lib.lib()
    res = out if out is not None else torch.empty(N, dtype=torch.float32, device=x.device)
    nbytes = L.salun_colsum_f32_workspace_bytes(c_int64(M), N)
    ws = ops.workspace(nbytes, x.device)
    check(L.salun_colsum_f32(c_void_p(x.data_ptr()), c_void_p(res.data_ptr()), c_int64(M), N, c_int64(x.stride(0)),
                             int(bool(accumulate and out is not None)), c_void_p(ws.data_ptr()), c_size_t(ws.numel()),
                             c_void_p(torch.cuda.current_stream().cuda_stream)), "salun_colsum_f32")
    return res


def softmax_rows_(s: torch.Tensor) -> torch.Tensor:
    """In place over the last dimension of a contiguous fp32 tensor."""
    _f32_dev(s, "s")
    assert s.is_contiguous()
    n = s.shape[-1]
    check(_lib.lib().salun_softmax_rows(c_void_p(s.data_ptr()), c_int64(s.numel() // n), n, n,
                                        c_void_p(torch.cuda.current_stream().cuda_stream)), "salun_softmax_rows")
    return s


def softmax_rows_backward_(p: torch.Tensor, dp: torch.Tensor, scale: float) -> torch.Tensor:
    assert p.is_contiguous() and dp.is_contiguous() and p.shape == dp.shape
    n = p.shape[-1]
    check(_lib.lib().salun_softmax_rows_backward(c_void_p(p.data_ptr()), c_void_p(dp.data_ptr()),
                                                 c_int64(p.numel() // n), n, n, c_double(scale),
                                                 c_void_p(torch.cuda.current_stream().cuda_stream)),
          "salun_softmax_rows_backward")
    return dp


# ----------------------------------------------------------------------------------------------- Linear
# Where K15 is used.  It is a compact kernel (64 / 128-square tiles, 16-deep register-staged chunks) whose value is what it
# fuses — 22 embedding projections in one launch, an attention whose operands stay in the caller's layout, gradients added
# straight into the flat arena — on the small and medium problems of the DDPM U-Net.  On the plain multi-GFLOP products of
# the fp32 SD configuration the library GEMM is 1.5x faster (profiles/r04_gemmbench_f32.txt: 83-87 vs 128-140 TFLOP/s, the
# weight-gradient orientation 37-62 vs 55-116; whole step 1070 ms on K15 vs 765 ms), so those stay library GEMMs, and an
# attention whose score matrix passes 512 x 512 per head goes to the library's fused attention instead of materialising it.
LINEAR_MAX_FLOP = float(os.environ.get("SALUN_K15_MAX_GFLOP", "1.0")) * 1e9
ATTENTION_MAX_SCORES = 512 * 512


def _eligible(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_cuda
            and not torch.is_autocast_enabled() and x.shape[-1] == weight.shape[1] and x.numel() > 0
            and 2.0 * x.numel() * weight.shape[0] <= LINEAR_MAX_FLOP)


class _Linear(FastFunction):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        y = mm_nt(x2, weight, bias=bias)
        ctx.save_for_backward(x2, weight, bias)
        ctx.params = (weight, bias)  # the Parameter objects (gradient destinations): saved_tensors may be detached aliases
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        N, K = weight.shape
        dy2 = dy.reshape(-1, N)
        if dy2.stride(1) != 1:
            dy2 = dy2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX[m, k] = sum_n dY[m, n] W[n, k]
            dx = torch.empty_like(x2)
            launch([Job(dx.data_ptr(), x2.shape[0], K, dx.stride(0),
                        [Seg(dy2.data_ptr(), dy2.stride(0), 1, weight.data_ptr(), 1, weight.stride(0), N)])], dy.device)
            dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            # dW[n, k] = sum_m dY[m, n] x[m, k], added straight into the flat gradient when it can be
            sink = gradsink.sink(ctx.params[0])
            dst = sink if sink is not None else torch.empty_like(weight)
            launch([Job(dst.data_ptr(), N, K, dst.stride(0),
                        [Seg(dy2.data_ptr(), 1, dy2.stride(0), x2.data_ptr(), 1, x2.stride(0), x2.shape[0])],
                        accumulate=sink is not None)], dy.device)
            dw = None if sink is not None else dst
        if bias is not None and ctx.needs_input_grad[2]:
            sink = gradsink.sink(ctx.params[1])
            db = colsum(dy2, out=sink, accumulate=sink is not None)
            if sink is not None:
                db = None
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    if not _eligible(x, weight):
        return F.linear(x, weight, bias)
    return _Linear.apply(x, weight, bias)


class SalunLinear(nn.Linear):
    """nn.Linear whose fp32 device path is K15 (same parameters, same state_dict)."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


def use_salun_linears(model: nn.Module, skip: Sequence[str] = ()) -> int:
    """Re-class every plain fp32 nn.Linear of `model` (except module names containing an entry of `skip`)."""
    n = 0
    for name, mod in model.named_modules():
        if type(mod) is nn.Linear and mod.weight.dtype == torch.float32 and not any(s in name for s in skip):
            mod.__class__ = SalunLinear
            n += 1
    return n


# ------------------------------------------------------------------------ grouped Linear (shared input)
class _GroupedLinear(FastFunction):
    """y_g = x W_g^T + b_g for g = 0..G-1 with ONE input x [M, K]."""

    @staticmethod
    def forward(ctx, x, *wb):
        G = len(wb) // 2
        ws, bs = wb[:G], wb[G:]
        x2 = x if x.stride(1) == 1 else x.contiguous()
        M, K = x2.shape
        outs = [torch.empty((M, w.shape[0]), dtype=torch.float32, device=x.device) for w in ws]
        for g0 in range(0, G, _lib.SALUN_GEMM_MAX_JOBS):
            jobs = [Job(o.data_ptr(), M, w.shape[0], o.stride(0),
                        [Seg(x2.data_ptr(), x2.stride(0), 1, w.data_ptr(), w.stride(0), 1, K)],
                        b.data_ptr() if b is not None else 0)
                    for o, w, b in zip(outs[g0:g0 + 32], ws[g0:g0 + 32], bs[g0:g0 + 32])]
            launch(jobs, x.device)
        ctx.save_for_backward(x2, *ws)
        ctx.weights = ws  # Parameter objects (gradient destinations)
        ctx.biases = bs
        ctx.G = G
        ctx.set_materialize_grads(False)  # a projection whose block saw no gradient arrives as None, not as zeros
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x2, *ws = ctx.saved_tensors
        G, bs = ctx.G, ctx.biases
        M, K = x2.shape
        live = [(g, (dy if dy.stride(1) == 1 else dy.contiguous())) for g, dy in enumerate(dys) if dy is not None]
        dx = None
        if ctx.needs_input_grad[0] and live:
            # dX = sum_g dY_g W_g: ONE job whose reduction chains the groups
            dx = torch.empty_like(x2)
            for c0 in range(0, len(live), _lib.SALUN_GEMM_MAX_SEGS):
                part = live[c0:c0 + _lib.SALUN_GEMM_MAX_SEGS]
                segs = [Seg(dy.data_ptr(), dy.stride(0), 1, ws[g].data_ptr(), 1, ws[g].stride(0), ws[g].shape[0])
                        for g, dy in part]
                launch([Job(dx.data_ptr(), M, K, dx.stride(0), segs, accumulate=c0 > 0)], x2.device)
        dws: List[Optional[torch.Tensor]] = [None] * G
        dbs: List[Optional[torch.Tensor]] = [None] * G
        jobs, keep = [], []
        for g, dy in live:
            if not ctx.needs_input_grad[1 + g]:
                continue
            sink = gradsink.sink(ctx.weights[g])
            dst = sink if sink is not None else torch.empty_like(ws[g])
            keep.append(dst)
            jobs.append(Job(dst.data_ptr(), ws[g].shape[0], K, dst.stride(0),
                            [Seg(dy.data_ptr(), 1, dy.stride(0), x2.data_ptr(), 1, x2.stride(0), M)],
                            accumulate=sink is not None))
            if sink is None:
                dws[g] = dst
        for j0 in range(0, len(jobs), _lib.SALUN_GEMM_MAX_JOBS):
            launch(jobs[j0:j0 + _lib.SALUN_GEMM_MAX_JOBS], x2.device)
        for g, dy in live:
            b = bs[g]
            if b is None or not ctx.needs_input_grad[1 + G + g]:
                continue
            sink = gradsink.sink(b)
            r = colsum(dy, out=sink, accumulate=sink is not None)
            if sink is None:
                dbs[g] = r
        return (dx, *dws, *dbs)


def grouped_linear(x: torch.Tensor, layers: Sequence[nn.Linear]) -> Tuple[torch.Tensor, ...]:
    """Tuple of `layer(x)` for Linear layers that all read the same 2-D fp32 device input — one launch per direction."""
    ws = [l.weight for l in layers]
    bs = [l.bias for l in layers]
    if (x.dim() != 2 or not all(_eligible(x, w) for w in ws) or any(b is None for b in bs)):
        return tuple(l(x) for l in layers)
    return _GroupedLinear.apply(x, *ws, *bs)


# ----------------------------------------------------------------------------------------------- attention
def _bmm(c: torch.Tensor, ci: int, cj: int, a: torch.Tensor, ai: int, ak: int, b: torch.Tensor, bj: int, bk: int,
         alpha: float = 1.0) -> None:
    """c[B, H][i, j] = alpha * sum_k a[B, H][i, k] b[B, H][j, k] over 4-D views [B, H, ., .]: `ci / cj`, `ai / ak`,
    `bj / bk` name which of the last two dimensions (2 or 3) plays the role.  Whichever of c's two strides is 1 becomes
    the kernel's column index (the problem is transposed if needed)."""
    B, H = c.shape[0], c.shape[1]
    M, N, K = c.shape[ci], c.shape[cj], a.shape[ak]
    assert a.shape[ai] == M and b.shape[bj] == N and b.shape[bk] == K, (c.shape, a.shape, b.shape)
    if c.stride(cj) == 1 or N == 1:
        job = Job(c.data_ptr(), M, N, c.stride(ci),
                  [Seg(a.data_ptr(), a.stride(ai), a.stride(ak), b.data_ptr(), b.stride(bj), b.stride(bk), K)])
        strides = (a.stride(0), a.stride(1), b.stride(0), b.stride(1), c.stride(0), c.stride(1))
    elif c.stride(ci) == 1 or M == 1:  # C^T = B A^T
        job = Job(c.data_ptr(), N, M, c.stride(cj),
                  [Seg(b.data_ptr(), b.stride(bj), b.stride(bk), a.data_ptr(), a.stride(ai), a.stride(ak), K)])
        strides = (b.stride(0), b.stride(1), a.stride(0), a.stride(1), c.stride(0), c.stride(1))
    else:
        raise ValueError("the output needs one unit-stride dimension")
    launch([job], c.device, alpha, (B, H), strides)


def _dense_like(t: torch.Tensor) -> torch.Tensor:
    """An uninitialised tensor with t's shape AND strides when t is dense (so outputs land in the caller's layout)."""
    dims = sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1)
    expect, dense = 1, True
    for st, sz in dims:  # non-overlapping and dense: sorted by stride, each stride is the extent of everything below it
        dense = dense and st == expect
        expect *= sz
    if dense:
        return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
    return torch.empty(t.shape, dtype=t.dtype, device=t.device)


class _AttentionF32(FastFunction):
    """q [B, H, Tq, D], k / v [B, H, Tk, D] (views of any strides with one unit-stride dimension among the last two)."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        B, H, Tq, D = q.shape
        Tk = k.shape[2]
        p = torch.empty((B, H, Tq, Tk), dtype=torch.float32, device=q.device)
        _bmm(p, 2, 3, q, 2, 3, k, 2, 3, alpha=scale)        # S = scale q k^T
        softmax_rows_(p)
        o = _dense_like(q)
        _bmm(o, 2, 3, p, 2, 3, v, 3, 2)                     # O[i, d] = sum_j P[i, j] v[j, d]
        ctx.save_for_backward(q, k, v, p)
        ctx.scale = float(scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        if do.stride(2) != 1 and do.stride(3) != 1:
            do = do.contiguous()
        dv = _dense_like(v)
        _bmm(dv, 2, 3, p, 3, 2, do, 3, 2)                   # dV[j, d] = sum_i P[i, j] dO[i, d]
        dp = torch.empty_like(p)
        _bmm(dp, 2, 3, do, 2, 3, v, 2, 3)                   # dP[i, j] = sum_d dO[i, d] v[j, d]
        softmax_rows_backward_(p, dp, ctx.scale)            # dS (scale folded in)
        dq = _dense_like(q)
        _bmm(dq, 2, 3, dp, 2, 3, k, 3, 2)                   # dQ[i, d] = sum_j dS[i, j] k[j, d]
        dk = _dense_like(k)
        _bmm(dk, 2, 3, dp, 3, 2, q, 3, 2)                   # dK[j, d] = sum_i dS[i, j] q[i, d]
        return dq, dk, dv, None


def attention_f32(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(scale q k^T) v for fp32 device views [B, H, T, D]; the result has q's memory layout."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _f32_dev(t, n)
        if t.dim() != 4 or (t.stride(2) != 1 and t.stride(3) != 1):
            raise ValueError(f"{n}: expected a 4-D view [B, H, T, D] with a unit stride among its last two dimensions")
    return _AttentionF32.apply(q, k, v, float(scale))


def attention_supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> bool:
    return (all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and (t.stride(2) == 1 or t.stride(3) == 1)
                for t in (q, k, v)) and not torch.is_autocast_enabled() and q.shape[0] * q.shape[1] <= 65535
            and q.shape[2] * k.shape[2] <= ATTENTION_MAX_SCORES)
