# Round 5, GPU call 1: the rewritten top-k route (parity + per-kernel times), the new default bench line, and the
# same-box A/B of the round-3 tree against this one (the in-step update-kernel drift).
#   build_lab/r03 = `git archive bff0ec6` + make (built in the CPU container; travels with the snapshot)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -k "topk" -x -q -s 2>&1 | grep -v "amdgpu.ids" | tail -25 ) > gpurun_out/r05_run1_tests.txt 2>&1
( timeout 600 python -m pytest tests/test_next_gpu.py -x -q 2>&1 | tail -5 ) >> gpurun_out/r05_run1_tests.txt 2>&1
for cfg in "n18 1" "n18 10" "nd 1" "nd 10" "ns 1"; do
  set -- $cfg
  timeout 300 python tools/topk_prof.py $1 $2 20 2>&1 | grep "mask_topk n="
done > gpurun_out/r05_run1_topk.txt 2>&1
for cfg in "n18 1" "n18 10" "nd 1"; do
  set -- $cfg
  KEEP_TRACE=0 timeout 300 bash tools/prof.sh r05a_topk_$1_$2 python tools/topk_prof.py $1 $2 10 2>&1 | grep "k_\|mask_topk" | cut -c1-150
done >> gpurun_out/r05_run1_topk.txt 2>&1
timeout 900 python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
for i in 1 2; do
  ( cd build_lab/r03 && timeout 300 python bench.py --steps 177 --warmup 10 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('r03tree', round(d['value'],2), round(d['ms_per_step'],3), round(d['fwd_bwd']['frac'],4), 'sgd_us', round(r['mean_launch_us'],2), round(r['frac'],4))" )
  timeout 300 python bench.py --steps 177 --warmup 10 --no_cpu_baseline --no_ddpm 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('r05tree', round(d['value'],2), round(d['ms_per_step'],3), round(d['fwd_bwd']['frac'],4), 'sgd_us', round(r['mean_launch_us'],2), round(r['frac'],4), 'pair', r['event_pair_overhead_us'], 'b2b', r['back_to_back_launch_us'], 'topk', d['mask_gen'])"
done > gpurun_out/r05_run1_ab.txt 2>&1
cat gpurun_out/r05_run1_tests.txt gpurun_out/r05_run1_topk.txt gpurun_out/r05_run1_ab.txt
tail -c 1500 gpurun_out/r05a_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05a_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["fwd_bwd"]["frac"], d["mask_gen"])
print("ddpm", {k: v for k, v in d.get("ddpm", {}).items() if k in ("value", "ms_per_step", "error")}, (d.get("ddpm") or {}).get("roofline"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cpu_model"))
PY
