"""CPU oracle for the SalUn hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the timed CPU baseline.  Nothing under
``unlearn_saliency_amd/`` imports it (tests/test_layout.py enforces that).

Two layers:

* ``salun_oracle.c`` (built to ``oracle/liboracle.so`` by ``make -C oracle`` or
  :func:`build`): scalar C restatement of the reference's element-wise arithmetic and
  of the double-argsort mask, each function citing the reference file:line it follows.
* numpy restatements in this file (``*_numpy``) that follow the reference even more
  literally (``np.argsort(kind="stable")`` twice) and are used to cross-check the C at
  small sizes.

Parity pinning: the reference ships no tests or vectors, so the oracle is pinned against
outputs of the reference's own functions imported in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_vs_golden.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile salun_oracle.c with gcc (seconds).  Returns the .so path."""
    src = os.path.join(_HERE, "salun_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_clip_coef.restype = ctypes.c_float
        L.oracle_clip_coef.argtypes = [ctypes.c_float, ctypes.c_float]
        L.oracle_grad_sqnorm.restype = ctypes.c_float
        L.oracle_grad_sqnorm.argtypes = [_f32p, ctypes.c_int64]
        L.oracle_mask_topk.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(a: np.ndarray) -> np.ndarray:
    assert a.dtype == np.float32 and a.flags.c_contiguous, (a.dtype, a.flags)
    return a


def _p(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


# ----------------------------------------------------------------------------- K2
def k_of(n: int, ratio: float) -> int:
    """threshold_index = int(len(all_elements) * i) — Classification/generate_mask.py:60."""
    return int(n * ratio)


def mask_topk(acc: np.ndarray, ks: Sequence[int]) -> list[np.ndarray]:
    """C oracle: one u8 mask per k (stable double-argsort reading of generate_mask.py:57-79)."""
    acc = _f32(np.ascontiguousarray(acc.reshape(-1)))
    n = acc.size
    masks = [np.empty(n, dtype=np.uint8) for _ in ks]
    karr = (ctypes.c_int64 * len(ks))(*[int(k) for k in ks])
    marr = (_u8p * len(ks))(*[_p(m, _u8p) for m in masks])
    rc = lib().oracle_mask_topk(_p(acc, _f32p), ctypes.c_int64(n), karr, ctypes.c_int(len(ks)), marr)
    if rc != 0:
        raise RuntimeError("oracle_mask_topk failed (n too large for the oracle?)")
    return masks


def mask_topk_numpy(acc: np.ndarray, ks: Sequence[int]) -> list[np.ndarray]:
    """Literal numpy restatement of Classification/generate_mask.py:46-79:

    abs -> negate -> argsort -> argsort -> ranks < k, with kind='stable' (NaN last)."""
    all_elements = -np.abs(acc.reshape(-1).astype(np.float32))
    positions = np.argsort(all_elements, kind="stable")
    ranks = np.argsort(positions, kind="stable")
    return [(ranks < int(k)).astype(np.uint8) for k in ks]


# -------------------------------------------------------------------- element-wise
def saliency_accumulate(acc: np.ndarray, g: np.ndarray, scale: float = 1.0) -> None:
    lib().oracle_saliency_accumulate(_p(_f32(acc), _f32p), _p(_f32(g), _f32p), ctypes.c_float(scale),
                                     ctypes.c_int64(acc.size))


def clip_coef(sqnorm: float, max_norm: float) -> float:
    return float(lib().oracle_clip_coef(ctypes.c_float(sqnorm), ctypes.c_float(max_norm)))


def grad_sqnorm(g: np.ndarray) -> float:
    return float(lib().oracle_grad_sqnorm(_p(_f32(g), _f32p), ctypes.c_int64(g.size)))


def masked_sgd_step(p, g, buf, m, lr, mu, wd, first_step) -> None:
    """Fused form (what the HIP kernel computes)."""
    lib().oracle_masked_sgd_step(_p(_f32(p), _f32p), _p(_f32(g), _f32p), _p(buf, _f32p), _p(m, _u8p),
                                 ctypes.c_double(lr), ctypes.c_double(mu), ctypes.c_double(wd),
                                 ctypes.c_int(int(first_step)), ctypes.c_int64(p.size))


def masked_sgd_step_reference(p, g, buf, m, theta0, lr, mu, wd, first_step) -> None:
    """Literal mask-multiply -> SGD -> restore sequence (RL.py:11-34 + impl.py:68-73)."""
    lib().oracle_masked_sgd_step_reference(_p(_f32(p), _f32p), _p(_f32(g), _f32p), _p(buf, _f32p), _p(m, _u8p),
                                           _p(theta0, _f32p), ctypes.c_double(lr), ctypes.c_double(mu),
                                           ctypes.c_double(wd), ctypes.c_int(int(first_step)),
                                           ctypes.c_int64(p.size))


def masked_adam_step(p, g, m1, v, mask, gscale, lr, b1, b2, eps, wd, step) -> None:
    lib().oracle_masked_adam_step(_p(_f32(p), _f32p), _p(_f32(g), _f32p), _p(_f32(m1), _f32p), _p(_f32(v), _f32p),
                                  _p(mask, _u8p), ctypes.c_double(gscale), ctypes.c_double(lr), ctypes.c_double(b1),
                                  ctypes.c_double(b2), ctypes.c_double(eps), ctypes.c_double(wd), ctypes.c_int(step),
                                  ctypes.c_int64(p.size))


def qsample(x0, e, sqrt_ab, sqrt_1mab, t) -> np.ndarray:
    B = x0.shape[0]
    chw = x0.size // B
    xt = np.empty_like(x0)
    t = np.ascontiguousarray(t, dtype=np.int64)
    lib().oracle_qsample(_p(_f32(x0), _f32p), _p(_f32(e), _f32p), _p(_f32(sqrt_ab), _f32p), _p(_f32(sqrt_1mab), _f32p),
                         _p(t, _i64p), ctypes.c_int64(sqrt_ab.size), _p(xt, _f32p), ctypes.c_int64(B),
                         ctypes.c_int64(chw))
    return xt


def sqerr_loss(a, b, coef, want_grad=True):
    """returns (loss, per_sample, dloss_db)."""
    B = a.shape[0]
    chw = a.size // B
    loss = np.zeros(1, np.float32)
    per = np.zeros(B, np.float32)
    d = np.empty_like(a) if want_grad else None
    lib().oracle_sqerr_loss(_p(_f32(a), _f32p), _p(_f32(b), _f32p), ctypes.c_int64(B), ctypes.c_int64(chw),
                            ctypes.c_double(coef), _p(loss, _f32p), _p(per, _f32p), _p(d, _f32p))
    return float(loss[0]), per, d


def fim_square_accumulate(F, tmp, n_data) -> None:
    lib().oracle_fim_square_accumulate(_p(_f32(F), _f32p), _p(_f32(tmp), _f32p), ctypes.c_double(n_data),
                                       ctypes.c_int64(F.size))


def image_batch(data, idx, crop, flip, pad) -> np.ndarray:
    num, H, W, C = data.shape
    B = len(idx)
    out = np.empty((B, C, H, W), np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    crop = None if crop is None else np.ascontiguousarray(crop, dtype=np.int32)
    flip = None if flip is None else np.ascontiguousarray(flip, dtype=np.uint8)
    assert data.dtype == np.uint8 and data.flags.c_contiguous
    lib().oracle_image_batch(_p(data, _u8p), _p(idx, _i64p), _p(crop, _i32p), _p(flip, _u8p), _p(out, _f32p),
                             ctypes.c_int64(B), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(C), ctypes.c_int(pad))
    return out


# --------------------------------------------------------------------- generators
def fill_uniform(n: int, seed: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().oracle_fill_uniform(_p(out, _f32p), ctypes.c_int64(n), ctypes.c_uint64(seed), ctypes.c_double(lo),
                              ctypes.c_double(hi))
    return out


def fill_normal(n: int, seed: int, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().oracle_fill_normal(_p(out, _f32p), ctypes.c_int64(n), ctypes.c_uint64(seed), ctypes.c_double(mean),
                             ctypes.c_double(std))
    return out


def fill_u8(n: int, seed: int) -> np.ndarray:
    out = np.empty(n, np.uint8)
    lib().oracle_fill_u8(_p(out, _u8p), ctypes.c_int64(n), ctypes.c_uint64(seed))
    return out


def dropout(x: np.ndarray, p: float, key: int, sample_offset: int = 0) -> np.ndarray:
    """Counter-based dropout of a (n_samples, ...) fp32 batch whose first sample has GLOBAL index `sample_offset`
    (oracle_dropout; the device kernel is salun_dropout).  The same call on dy is the backward pass."""
    x = _f32(np.ascontiguousarray(x))
    n = x.shape[0]
    chw = x.size // max(n, 1)
    y = np.empty_like(x)
    lib().oracle_dropout(_p(x, _f32p), _p(y, _f32p), ctypes.c_int64(n), ctypes.c_int64(chw),
                         ctypes.c_int64(sample_offset), ctypes.c_double(p), ctypes.c_uint64(key & 0xFFFFFFFFFFFFFFFF))
    return y


# ------------------------------------------------------------------ next rows, numpy
def proximal_threshold(p: np.ndarray, p0: np.ndarray, ratio: int) -> np.float32:
    """threshold = -torch.topk(-|p - p0|, ratio)[0][-1]  — the ratio-th smallest |p - p0|
    (Classification/unlearn/RL_pro.py:53-56).  ratio < 1 raises like the reference's [-1] on an empty result."""
    if ratio < 1:
        raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
    d = np.abs(_f32(p) - _f32(p0))
    return np.partition(d, ratio - 1)[ratio - 1]


def soft_threshold_step(p: np.ndarray, p0: np.ndarray, ratio: int) -> np.float32:
    """In place: params = where(d > thr, params - thr, where(d < -thr, params + thr, init_params))
    (RL_pro.py:56-58), all fp32.  Returns the threshold."""
    thr = np.float32(proximal_threshold(p, p0, ratio))
    d = _f32(p) - _f32(p0)
    p[...] = np.where(d > thr, p - thr, np.where(d < -thr, p + thr, p0)).astype(np.float32)
    return thr


def ewc_penalty_grad(p, p_star, F, g, lam: float):
    """EWC term of DDPM train_forget (DDPM/runners/diffusion.py:343-350): returns lam * sum F (p - p*)^2 (fp64
    accumulation of fp32 products) and adds its autograd gradient (lam*F) * (2*(p - p*)) to g in place, fp32."""
    lam32 = np.float32(lam)
    d = _f32(p) - _f32(p_star)
    s = float(np.sum((_f32(F) * (d * d)).astype(np.float64)))
    g += (lam32 * _f32(F)) * (np.float32(2.0) * d)
    return float(lam32) * s, s
