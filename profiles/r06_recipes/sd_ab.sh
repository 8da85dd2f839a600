# Round 6, the SD bf16 step: same-box A/B runs behind the numbers of DESIGN.md §6b "Round 6" (each pair alternated twice in
# one gpurun call; bench_sd.py prints value / ms_per_step / host_enqueue_ms_per_step).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() {  # one <label> [ENV=VALUE ...] -- [bench.py flags]
  label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --workload sd --steps 5 --warmup 2 --no_cpu_baseline "$@" > gpurun_out/sd_ab.json 2> gpurun_out/sd_ab.err
  python - "$label" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/sd_ab.json") if l.startswith("{")][-1]); r = d.get("resident_activations") or {}
print(sys.argv[1], round(d["value"], 3), round(d["ms_per_step"], 2), round(d["host_enqueue_ms_per_step"], 1),
      "| resident", r.get("value"), r.get("ms_per_step"), r.get("host_enqueue_ms_per_step"))
PY
}
for i in 1 2; do
  one "pack per layer     " SALUN_BF16_BATCH_PACK=0 --
  one "pack per model     " SALUN_BF16_BATCH_PACK=1 --
  one "emb under autocast " SALUN_SD_EMB_FP32=0 --
  one "emb fp32, one SiLU " SALUN_SD_EMB_FP32=1 --
  one "wgrad on main      " SALUN_BF16_WGRAD_OVERLAP=0 --
  one "wgrad beside       " SALUN_BF16_WGRAD_OVERLAP=1 --
done
# data parallel at world size 1 (RCCL): with / without the target pass on a stream of its own (the tree's rule: without)
export MASTER_PORT=29637
one "dp                 " A=1 -- --force_collectives
one "dp, 24 hw queues   " GPU_MAX_HW_QUEUES=24 -- --force_collectives
one "plain              " A=1 --
# diagnostics
python tools/bench_sd.py --bf16 --steps 2 --warmup 1 --no_cpu_baseline --aten_origins 2>&1 >/dev/null | grep aten_origins | head -40
python tools/hostprof_diffusion.py sd | head -60
