"""ctypes binding of libsalun.so — the C-ABI declared in include/salun.h.

The library is built in-tree (``make -C unlearn_saliency_amd/csrc`` or
``__graft_entry__.build()``) and loaded from this directory.  There is no CPU
fallback anywhere in this package: if the shared object is missing or a symbol is
absent, importing callers get a loud ``ImportError`` / ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SALUN_LIB") or os.path.join(_HERE, "libsalun.so")  # SALUN_LIB: A/B builds when tuning
CSRC_DIR = os.path.join(_HERE, "csrc")

SALUN_OK = 0
SALUN_EINVAL = -22
SALUN_ENOSPC = -28
SALUN_EIO = -5
SALUN_MAX_THRESHOLDS = 16
SALUN_TOPK_FORCE_FULL_SCAN = 1
SALUN_TOPK_VALUES_ONLY = 2
SALUN_WGRAD_SHARED = 1

c_void_p, c_int, c_int64, c_uint64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64
c_double, c_size_t = ctypes.c_double, ctypes.c_size_t

class GemmSeg(ctypes.Structure):
    """salun_gemm_seg_t"""
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("K", ctypes.c_int32), ("a_i", ctypes.c_int32),
                ("a_k", ctypes.c_int32), ("b_j", ctypes.c_int32), ("b_k", ctypes.c_int32)]


class GemmJob(ctypes.Structure):
    """salun_gemm_job_t"""
    _fields_ = [("C", c_void_p), ("bias", c_void_p), ("M", ctypes.c_int32), ("N", ctypes.c_int32),
                ("ldc", ctypes.c_int32), ("seg0", ctypes.c_int32), ("nseg", ctypes.c_int32),
                ("accumulate", ctypes.c_int32)]


class PackJob(ctypes.Structure):
    """salun_pack_job_t"""
    _fields_ = [("w", c_void_p), ("img_fwd", c_void_p), ("img_dgrad", c_void_p), ("K", ctypes.c_int32),
                ("C", ctypes.c_int32)]


class Bf16PackJob(ctypes.Structure):
    """salun_bf16_pack_job_t"""
    _fields_ = [("w", c_void_p), ("wp", c_void_p), ("K", ctypes.c_int32), ("C", ctypes.c_int32), ("R", ctypes.c_int32),
                ("transposed", ctypes.c_int32)]


SALUN_BF16_PACK_MAX_JOBS = 64
SALUN_GEMM_MAX_JOBS = 32
SALUN_GEMM_MAX_SEGS = 32

# name -> (restype, argtypes); mirrors include/salun.h one to one
# (tests/test_cabi.py checks this table against the header and the .so).
SIGNATURES = {
    "salun_version": (c_int, []),
    "salun_strerror": (ctypes.c_char_p, [c_int]),
    "salun_arch": (ctypes.c_char_p, []),
    "salun_clock_probe": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "salun_saliency_accumulate": (c_int, [c_void_p, c_void_p, c_double, c_void_p, c_double, c_int64, c_void_p]),
    "salun_mask_topk_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "salun_mask_topk": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_int64), c_int, ctypes.POINTER(c_void_p),
                                c_void_p, c_size_t, c_void_p]),
    "salun_mask_topk_ex": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_int64), c_int, ctypes.POINTER(c_void_p),
                                   c_void_p, c_size_t, ctypes.c_uint, c_void_p]),
    "salun_mask_topk_status": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p]),
    "salun_mask_topk_thresholds": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "salun_mask_u8_to_i64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "salun_mask_i64_to_u8": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "salun_mask_popcount": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "salun_masked_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_int,
                                      c_int64, c_void_p]),
    "salun_reduce_workspace_bytes": (c_size_t, [c_int64]),
    "salun_grad_sqnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "salun_masked_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                       c_double, c_double, c_double, c_double, c_double, c_double, c_int, c_int64,
                                       c_void_p]),
    "salun_adam_coefficients": (c_int, [c_void_p, c_double, c_double, c_double, c_void_p, c_void_p]),
    "salun_masked_adam_step_coef": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                            c_double, c_void_p, c_double, c_double, c_double, c_double, c_int64,
                                            c_void_p]),
    "salun_qsample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                              c_void_p]),
    "salun_sqerr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "salun_sqerr_loss": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "salun_fim_square_accumulate": (c_int, [c_void_p, c_void_p, c_double, c_int64, c_void_p]),
    "salun_conv2d_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "salun_conv2d_forward_fused": (c_int, [c_void_p] * 6 + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    "salun_conv2d_data_workspace_bytes": (c_size_t, [c_int] * 6),
    "salun_conv2d_backward_data_ws": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    "salun_conv2d_backward_data": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "salun_channel_sum_workspace_bytes": (c_size_t, [c_int] * 2),
    "salun_channel_sum": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    "salun_conv2d_wgrad_workspace_bytes": (c_size_t, [c_int] * 6),
    "salun_conv3x3_pack_bytes": (c_size_t, [c_int] * 3),
    "salun_conv3x3_pack_weights": (c_int, [c_void_p, c_int, c_void_p]),
    "salun_conv3x3_packed": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "salun_conv2d_backward_weight": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p, c_size_t,
                                                                                          c_void_p]),
    "salun_conv2d_backward_weight_ex": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [ctypes.c_uint, c_void_p,
                                                                                                 c_size_t, c_void_p]),
    "salun_conv2d_bf16_supported": (c_int, [c_int] * 5),
    "salun_conv2d_bf16_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "salun_bf16_pack_weights_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "salun_conv2d_bf16_data_workspace_bytes": (c_size_t, [c_int] * 8),
    "salun_conv2d_bf16_forward": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "salun_conv2d_bf16_backward_data": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "salun_conv2d_bf16_wgrad_workspace_bytes": (c_size_t, [c_int] * 8),
    "salun_conv2d_bf16_backward_weight": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]),
    "salun_colsum_bf16_workspace_bytes": (c_size_t, [c_int]),
    "salun_colsum_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "salun_conv2d_bf16_backward_weight_ex": (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]),
    "salun_gn_bf16_workspace_bytes": (c_size_t, [c_int] * 4),
    "salun_gn_bf16_forward": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_double, c_int, c_void_p, c_size_t, c_void_p]),
    "salun_gn_bf16_backward": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "salun_attn_supported": (c_int, [c_int]),
    "salun_attn_forward": (c_int, [c_void_p] * 5 + [c_int] * 5 + [ctypes.c_longlong, c_int] * 4 + [c_double, c_void_p]),
    "salun_attn_backward": (c_int, [c_void_p] * 10 + [c_int] * 5 + [ctypes.c_longlong, c_int] * 5 + [c_double, c_void_p]),
    "salun_ln_bf16_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "salun_ln_bf16_forward": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_double, c_void_p]),
    "salun_ln_bf16_backward": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "salun_geglu_bf16_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "salun_geglu_bf16_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "salun_bn_workspace_bytes": (c_size_t, [c_int]),
    "salun_bn_forward": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_double, c_double, c_int, c_void_p, c_size_t, c_void_p]),
    "salun_bn_backward": (c_int, [c_void_p] * 12 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "salun_conv2d_backward_data_add": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "salun_gn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "salun_gn_forward": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_double, c_int, c_void_p]),
    "salun_gn_backward": (c_int, [c_void_p] * 11 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "salun_gn_backward_fused": (c_int, [c_void_p] * 15 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "salun_param_diff": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "salun_soft_threshold_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "salun_ewc_workspace_bytes": (c_size_t, [c_int64]),
    "salun_ewc_penalty_grad": (c_int, [c_void_p] * 4 + [c_double, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "salun_image_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    "salun_gemm_f32_workspace_bytes": (c_size_t, [ctypes.POINTER(GemmJob), c_int, ctypes.POINTER(GemmSeg), c_int, c_int,
                                                  c_int]),
    "salun_gemm_f32": (c_int, [ctypes.POINTER(GemmJob), c_int, ctypes.POINTER(GemmSeg), c_int, c_int, c_int,
                               ctypes.POINTER(c_int64), c_double, c_void_p, c_size_t, c_void_p]),
    "salun_colsum_f32_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "salun_colsum_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "salun_softmax_rows": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p]),
    "salun_softmax_rows_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_double, c_void_p]),
    "salun_gemm_bf16_supported": (c_int, [c_int64, c_int, c_int]),
    "salun_gemm_bf16_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "salun_pack_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "salun_gemm_bf16_tn_supported": (c_int, [c_int64, c_int, c_int]),
    "salun_gemm_bf16_tn_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "salun_gemm_bf16_tn": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "salun_dropout": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_double, c_uint64, c_void_p, c_void_p]),
    "salun_u64_add": (c_int, [c_void_p, c_uint64, c_void_p]),
    "salun_fill_uniform": (c_int, [c_void_p, c_int64, c_uint64, c_double, c_double, c_void_p]),
    "salun_fill_normal": (c_int, [c_void_p, c_int64, c_uint64, c_double, c_double, c_void_p]),
    "salun_fill_u8": (c_int, [c_void_p, c_int64, c_uint64, c_void_p]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libsalun.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j4"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library with argtypes set.  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the SalUn HIP extension has not been built "
                f"(run `make -C {CSRC_DIR}` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:  # stale .so
                raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class SalunError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != SALUN_OK:
        msg = lib().salun_strerror(code).decode()
        raise SalunError(f"{what} failed: {msg} ({code})")
