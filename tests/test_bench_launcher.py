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


def _json_line(r):
    assert r.returncode == 0, r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_class_forget_workload_shards_over_two_ranks():
    """BASELINE configs[2] under the launcher: `bench.py --gpus 2 --forget class` — both ranks mark the same 4,500
    samples of class 0 and their shards partition every global batch (gloo on the CPU; the device step itself is the
    single-GPU path of tests/test_classwise_gpu.py plus the all-reduce of tests/test_dist_gloo.py)."""
    out = _json_line(_run(["--gpus", "2", "--forget", "class", "--selftest_workload"], {"OMP_NUM_THREADS": "1"}))
    assert out["n_gpus"] == 2 and out["forget"] == "class" and out["forget_classes"] == [0]
    assert out["forget_samples"] == 4500 and out["retain_samples"] == 40500
    assert out["global_batch"] == 512 and out["forget_batches"] == 9     # weak scaling: 256 per rank
    assert out["shards_partition_every_batch"] and out["shard_imbalance_max"] <= 1
    assert sum(out["tail_batch"]) == 4500 - 8 * 512


def test_strong_scaling_keeps_the_reference_global_batch():
    out = _json_line(_run(["--gpus", "2", "--scaling", "strong", "--selftest_workload"], {"OMP_NUM_THREADS": "1"}))
    assert out["forget"] == "random" and len(out["forget_classes"]) == 10
    assert out["global_batch"] == 256 and out["forget_batches"] == 18   # the reference's 18 forget batches per epoch
    assert out["shards_partition_every_batch"] and out["tail_batch"] == [74, 74]


def test_outlier_data_parallel_reading_is_measured_twice(monkeypatch):
    """bench.dp_ws1_checked: a reading more than 15 % above the single-process step is repeated; both stay in the block."""
    import importlib, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    readings = iter([{"ms_per_step": 12.9, "dp_over_plain": 1.5}, {"ms_per_step": 8.7, "dp_over_plain": 1.01}])
    monkeypatch.setattr(bench, "dp_ws1_line", lambda extra, plain: dict(next(readings)))
    r = bench.dp_ws1_checked([], 8.6)
    assert r["ms_per_step"] == 8.7 and r["attempts_ms"] == [12.9, 8.7] and "note" in r
    calls = []
    monkeypatch.setattr(bench, "dp_ws1_line", lambda extra, plain: calls.append(1) or {"ms_per_step": 8.8, "dp_over_plain": 1.02})
    r = bench.dp_ws1_checked([], 8.6)
    assert len(calls) == 1 and "attempts_ms" not in r
    monkeypatch.setattr(bench, "dp_ws1_line", lambda extra, plain: {"error": "x"})
    assert bench.dp_ws1_checked([], 8.6) == {"error": "x"}
