"""De-risking the RCCL branch on a single-GPU box.

Every data-parallel test in this suite forces gloo (two ranks cannot share one device under RCCL), so the `nccl`
branch — `init_process_group(device_id=...)`, `ReduceOp.AVG`, the async gradient buckets and their `work.wait()`
stream semantics, the immediate side-stream join of the wgrad kernels — had never executed anywhere.  With
SALUN_FORCE_COLLECTIVES=1 the package creates the process group and issues every collective at world size 1, over RCCL.
The result must be bit-identical to the single-process run (AVG over one rank is the identity; the kernels are the
same; only stream placement and the collectives differ)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script_args, env_extra, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
              "SALUN_DIST_BACKEND", "SALUN_FORCE_COLLECTIVES"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable] + script_args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_fused_steps_through_rccl_at_world_size_one_are_bit_identical_to_single_process():
    worker = os.path.join(ROOT, "tests", "_rccl_ws1_worker.py")
    plain = _run([worker], {})
    forced = _run([worker], {"SALUN_FORCE_COLLECTIVES": "1", "MASTER_PORT": "29611"})
    assert plain["backend"] is None and plain["collectives"] is False
    assert forced["backend"] == "nccl" and forced["collectives"] is True and forced["rccl_ranks"] == 1
    assert forced["buckets_launched_in_backward"] > 0       # the async AVG slices really went out during backward
    assert forced["sum_ok"] and forced["avg_ok"]
    assert forced["losses"] == plain["losses"], (forced["losses"], plain["losses"])
    assert forced["digest"] == plain["digest"]


def test_bench_line_through_rccl_at_world_size_one():
    out = _run([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force_collectives", "--steps", "12", "--warmup", "3",
                "--no_cpu_baseline"], {"MASTER_PORT": "29612"})
    assert out["backend"] == "nccl" and out["collectives"] is True and out["rccl_ranks"] == 1 and out["n_gpus"] == 1
    assert out["value"] > 0 and out["config"]["library_conv_calls"]["total"] == 0
    assert out["roofline"]["kernel"].endswith("(+ flat-gradient all-reduce)")
    assert out["samples_in_window"] == 12 * 256


@pytest.mark.parametrize("workload,steps", [("ddpm", 3), ("sd", 2)])
def test_diffusion_bench_through_rccl_at_world_size_one_is_bit_identical(workload, steps):
    """`bench.py --workload ddpm|sd --force_collectives`: the per-batch flat SUM of Phase A (DDPM: before the global
    clip), the async AVG gradient buckets behind the ResnetBlock / checkpointed nodes, the target pass on its own
    stream — all through RCCL at world size 1 — must leave the same parameters and the same last loss as the
    single-process run (reference being replaced: nn.DataParallel, DDPM/runners/diffusion.py:504,582-593,948-996)."""
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", workload, "--steps", str(steps), "--warmup", "1",
            "--no_cpu_baseline", "--digest"]
    plain = _run(args, {}, timeout=1500)
    forced = _run(args + ["--force_collectives"], {"MASTER_PORT": "29613"}, timeout=1500)
    assert plain["backend"] == "single-process" and plain["collectives"] is False
    assert forced["backend"] == "nccl" and forced["collectives"] is True and forced["rccl_ranks"] == 1
    assert forced["last_loss"] == plain["last_loss"], (forced["last_loss"], plain["last_loss"])
    assert forced["params_sha256"] == plain["params_sha256"]
    assert forced["value"] > 0 and forced["n_gpus"] == 1
