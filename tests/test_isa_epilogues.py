"""Host-only regression guard for the round-5 epilogue fix: no K11 kernel may go back to one `load -> s_waitcnt vmcnt(0)
-> store` per element (tools/isa_serial_loads.py counts the lone load/wait pairs in the gfx950 ISA; hipcc cross-compiles
without a GPU).  Before the fix conv_bf16_igemm had 129 such pairs and conv_bf16_wgrad<3, 1> 144."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_k11_kernels_have_no_serial_epilogue_loads():
    src = os.path.join(ROOT, "unlearn_saliency_amd", "csrc", "salun_conv_bf16.hip")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_serial_loads.py"), src], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [ln.split(None, 2) for ln in out.stdout.splitlines() if ln.strip() and ln.split()[0].isdigit()]
    worst = max((int(r[0]) for r in rows), default=0)
    assert worst < 16, out.stdout


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_k16_kernels_on_the_default_route_have_no_serial_epilogue_loads():
    """K16's ring kernels, the ring forward convolution and the TN weight-gradient GEMM (the ones the default dispatch
    picks; the numbered lab variants k_gemm_bf16_nt / _nt_p keep their row-by-row epilogues)."""
    src = os.path.join(ROOT, "unlearn_saliency_amd", "csrc", "salun_gemm.hip")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_serial_loads.py"), src], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    for ln in out.stdout.splitlines():
        f = ln.split(None, 2)
        if len(f) == 3 and f[0].isdigit() and any(k in f[2] for k in ("k_gemm_bf16_nt_r<", "k_conv_bf16_ring<", "k_gemm_bf16_tn<")):
            assert int(f[0]) < 16, ln
