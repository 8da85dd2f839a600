"""How much of each tolerance do the GPU parity tests actually use?  Runs the given test files in-process with
torch.allclose / np.allclose wrapped: for every call site (file:line) the worst  |a - b| / (atol + rtol |b|)  over all
calls is recorded (1.0 = the assertion is at its limit).  Tolerances are then set to ~3x what was measured on the MI355X
(VERDICT r3 weak #1).   python tools/measure_tolerances.py tests/test_next_gpu.py tests/test_conv_gpu.py ..."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pytest
import torch

use = {}


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/tests/" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def record(a, b, rtol, atol):
    a64, b64 = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = atol + rtol * np.abs(b64)
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = np.abs(a64 - b64) / den
    frac = float(np.nanmax(np.where(np.isfinite(frac), frac, 0.0))) if frac.size else 0.0
    k = site()
    use[k] = max(use.get(k, 0.0), frac)


real_t, real_n = torch.allclose, np.allclose


def t_allclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False):
    try:
        record(a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy(), rtol, atol)
    except Exception:
        pass
    return real_t(a, b, rtol=rtol, atol=atol, equal_nan=equal_nan)


def n_allclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False):
    try:
        record(a, b, rtol, atol)
    except Exception:
        pass
    return real_n(a, b, rtol=rtol, atol=atol, equal_nan=equal_nan)


torch.allclose, np.allclose = t_allclose, n_allclose
rc = pytest.main(["-q", "-p", "no:cacheprovider", "-s"] + sys.argv[1:])
out = os.path.join(ROOT, "gpurun_out", "r04_tolerance_use.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    for k, v in sorted(use.items()):
        f.write(f"{k} {v:.4f}\n")
print("pytest rc", rc, "->", out)
