"""The C-ABI boundary: libsalun.so builds for gfx950, loads without a GPU, and exports exactly the
entry points include/salun.h declares, with the arity the ctypes table binds.  No compute calls here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "salun.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # strip comments
    protos = re.findall(r"\b(?:int|size_t|const char \*)\s*(salun_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    out = {}
    for name, args in protos:
        args = args.strip()
        out[name] = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
    return out


@pytest.fixture(scope="module")
def built_lib():
    from unlearn_saliency_amd import _lib
    _lib.build()
    return _lib


def test_header_declares_the_hot_path_entry_points():
    fns = header_functions()
    for must in ("salun_saliency_accumulate", "salun_mask_topk", "salun_mask_topk_workspace_bytes",
                 "salun_mask_u8_to_i64", "salun_mask_i64_to_u8", "salun_masked_sgd_step", "salun_grad_sqnorm",
                 "salun_masked_adam_step", "salun_qsample", "salun_sqerr_loss", "salun_fim_square_accumulate",
                 "salun_image_batch"):
        assert must in fns, must


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    declared = set(header_functions())
    assert declared <= exported, declared - exported
    # nothing undeclared leaks out of the boundary
    assert {s for s in exported if s.startswith("salun_")} == declared


def test_ctypes_table_matches_header(built_lib):
    fns = header_functions()
    assert set(built_lib.SIGNATURES) == set(fns)
    for name, (_, argtypes) in built_lib.SIGNATURES.items():
        assert len(argtypes) == fns[name], name


def test_library_loads_without_gpu_and_identifies_itself(built_lib):
    L = built_lib.lib()
    assert L.salun_version() >= 100
    assert L.salun_arch() == b"gfx950"
    assert L.salun_strerror(0) == b"ok" and b"invalid" in L.salun_strerror(-22)
    assert L.salun_mask_topk_workspace_bytes(11_173_962, 10) > 272 * 1024
    assert L.salun_mask_topk_workspace_bytes(10, 17) == 0  # nk out of range


def test_code_object_targets_gfx950_only(built_lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", built_lib.LIB_PATH],
                         capture_output=True, text=True).stdout
    blob = open(built_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_90", b"gfx1100"):
        assert other not in blob, other


def test_missing_library_fails_loudly(monkeypatch, built_lib):
    monkeypatch.setattr(built_lib, "_lib", None)
    monkeypatch.setattr(built_lib, "LIB_PATH", "/nonexistent/libsalun.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        built_lib.lib()


def test_ops_reject_cpu_tensors():
    import torch
    from unlearn_saliency_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.masked_sgd_step(torch.zeros(8), torch.zeros(8), torch.zeros(8), None, 0.1, 0.9, 0.0, True)


def test_data_workspace_query_is_host_only_and_announces_the_reduction_split(built_lib):
    """`salun_conv2d_data_workspace_bytes` (host arithmetic, no device call): > 0 exactly for launches of <= 256
    workgroups — the DDPM U-Net's 4x4 level at batch 128 — and 0 for full launches, shapes outside the tiling
    domain and arguments that make no sense."""
    from unlearn_saliency_amd import _lib
    L = _lib.lib()
    q = L.salun_conv2d_data_workspace_bytes
    assert q(128, 256, 4, 4, 3, 1) == 8 * 128 * 256 * 16 * 4       # 32 tiles x 4 channel blocks = 128 workgroups
    assert q(128, 256, 4, 4, 1, 1) == 8 * 128 * 256 * 16 * 4
    assert q(128, 256, 8, 8, 3, 1) == 0                              # 512 workgroups: a full launch
    assert q(256, 64, 32, 32, 3, 1) == 0 and q(256, 512, 4, 4, 3, 1) == 0   # every ResNet-18 layer at batch 256
    assert q(4, 64, 6, 6, 3, 1) == 0                                 # width not a power of two: outside the tiling
    assert q(0, 64, 4, 4, 3, 1) == 0 and q(4, 64, 4, 4, 5, 1) == 0 and q(4, 64, 4, 4, 3, 3) == 0


def test_k16_domain_and_workspace_queries_are_host_only(built_lib):
    """`salun_gemm_bf16_supported`, `salun_gemm_bf16_tn_supported`, `salun_gemm_bf16_tn_workspace_bytes` and the 1x1 route of
    `salun_conv2d_bf16_wgrad_workspace_bytes` are host arithmetic (no device call): the domains the headers state, and a
    reduction split that writes whole fp32 images of the weight gradient."""
    from ctypes import c_int64
    from unlearn_saliency_amd import _lib
    L = _lib.lib()
    nt, tn, tnws = L.salun_gemm_bf16_supported, L.salun_gemm_bf16_tn_supported, L.salun_gemm_bf16_tn_workspace_bytes
    assert nt(c_int64(32768), 320, 320) and nt(c_int64(1), 64, 64)
    assert not nt(c_int64(128), 96, 64) and not nt(c_int64(128), 64, 40) and not nt(c_int64(0), 64, 64)
    assert tn(c_int64(616), 320, 768) and tn(c_int64(1), 32, 32)
    assert not tn(c_int64(616), 40, 64) and not tn(c_int64(616), 64, 100) and not tn(c_int64(1 << 24), 64, 64)
    for M, Na, Nb in ((32768, 320, 320), (2048, 1280, 1280), (512, 1280, 5120), (64, 32, 32)):
        b = tnws(c_int64(M), Na, Nb, 0)
        assert b % (Na * Nb * 4) == 0                      # 0 (one split adds into dw itself) or `splits` fp32 images
        assert b // (Na * Nb * 4) <= max(1, M // 256)      # a split reduces over >= 256 tokens
    assert tnws(c_int64(64), 32, 32, 0) == 0 and tnws(c_int64(32768), 320, 320, 0) > 0
    assert tnws(c_int64(616), 40, 64, 0) == 0              # outside the domain
    # a 1x1 / stride-1 / unpadded weight gradient with >= 1024 pixels is the TN GEMM: its partials + the column-sum slab
    w = L.salun_conv2d_bf16_wgrad_workspace_bytes
    gemm = tnws(c_int64(8 * 64 * 64), 320, 320, 0)
    assert w(8, 64, 64, 320, 320, 1, 1, 0) >= gemm > 0
    assert w(8, 64, 64, 320, 320, 3, 1, 1) > 0 and w(8, 64, 64, 4, 320, 3, 1, 1) == 0


def test_topk_workspace_query_is_host_only_and_sized_for_real_accumulators(built_lib):
    """`salun_mask_topk_workspace_bytes` is host arithmetic (no device call).  Its sizes follow the single-read route's
    layout: slabs of twice the mean candidate share per workgroup, a spill row and short-list segments for half of all
    candidates of a threshold — so they grow about linearly in nk, stay a small multiple of the vector for one threshold,
    and the arguments the C-ABI refuses give 0."""
    from ctypes import c_int64
    from unlearn_saliency_amd import _lib
    q = lambda n, nk: _lib.lib().salun_mask_topk_workspace_bytes(c_int64(n), nk)
    N18, ND, NS = 11_173_962, 38_632_323, 859_520_964
    assert q(0, 1) > 0 and q(1, 1) > 0 and q(8191, 1) > 0            # the full scan's state alone
    assert q(-1, 1) == 0 and q(N18, 0) == 0 and q(N18, _lib.SALUN_MAX_THRESHOLDS + 1) == 0
    for n in (8192, N18, ND, NS):
        one = q(n, 1)
        assert one < 4 * n + (4 << 20)                                # one threshold: well under the vector itself
        assert q(n, 2) > one
    assert q(N18, 10) < 16 * q(N18, 1) and q(ND, 10) < 16 * q(ND, 1)
    # the reference's calls: ten ratios on a ResNet-18, one ratio on the diffusion U-Nets
    assert q(N18, 10) < 256 << 20 and q(ND, 1) < 64 << 20 and q(NS, 1) < 256 << 20
    # 2^27 is where the brackets switch to the 2^20-element sample (a ~8x narrower bracket): the workspace of ONE
    # threshold drops across that boundary although the vector grows
    assert q((1 << 27) + 8, 1) < q((1 << 27) - 8, 1)
