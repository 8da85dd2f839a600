"""`python bench.py --gpus N` creates N ranks itself.

CPU-side: the launcher path is the same one the GPU run takes (`dist.launch_ranks` -> `torch.distributed.run` ->
`dist.init_from_env`); here the ranks rendezvous over gloo and only the self-test is run, no workload."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_gpus_2_selftest_creates_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--selftest_launcher"], {"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo", out


def test_gpus_2_without_two_devices_fails_loudly():
    """On a box with fewer than 2 devices (this container has none) the real benchmark refuses instead of silently
    running one rank."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two devices")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no_cpu_baseline"])
    assert r.returncode != 0
    assert "needs 2 devices" in (r.stderr + r.stdout), (r.stdout, r.stderr[-1500:])


def test_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "1", "--selftest_launcher"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "disagrees" in (r.stderr + r.stdout)
