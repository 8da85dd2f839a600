# Round 5, GPU call 2: top-k after the spill row / speculative loads / third level (all top-k tests incl. the bench
# accumulator), per-kernel times, and the SGD kernel's duration under rocprofv3 in the round-3 tree and in this one on
# the SAME box (what the HIP-event pair adds to it).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_kernels_gpu.py -k "topk" -q -s 2>&1 | grep -v "amdgpu.ids" | tail -40 ) > gpurun_out/r05_run2_tests.txt 2>&1
( timeout 600 python -m pytest tests/test_next_gpu.py -x -q 2>&1 | tail -5 ) >> gpurun_out/r05_run2_tests.txt 2>&1
for cfg in "n18 1" "n18 10" "nd 1" "nd 10" "ns 1"; do
  set -- $cfg
  timeout 300 python tools/topk_prof.py $1 $2 20 2>&1 | grep "mask_topk n="
done > gpurun_out/r05_run2_topk.txt 2>&1
for cfg in "n18 1" "n18 10" "nd 1" "ns 1"; do
  set -- $cfg
  KEEP_TRACE=0 timeout 300 bash tools/prof.sh r05b_topk_$1_$2 python tools/topk_prof.py $1 $2 10 > /dev/null 2>&1
done
timeout 600 python bench.py --no_cpu_baseline --no_ddpm > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err
( cd build_lab/r03 && KEEP_TRACE=0 timeout 300 bash $GRAFT_REPO_ROOT/tools/prof.sh r05b_r03tree python bench.py --steps 177 --warmup 10 --no_cpu_baseline > /dev/null 2>&1; cp gpurun_out/r05b_r03tree_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/ )
KEEP_TRACE=0 timeout 300 bash tools/prof.sh r05b_r05tree python bench.py --steps 177 --warmup 10 --no_cpu_baseline --no_ddpm > /dev/null 2>&1
cat gpurun_out/r05_run2_tests.txt gpurun_out/r05_run2_topk.txt
python - <<'PY'
import csv, json, glob
for f in sorted(glob.glob("gpurun_out/r05b_topk_*_kernel_stats.csv")):
    print("==", f)
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "k_" in n and not any(s in n for s in ("fill", "popcount", "partials")):
            print(f"  {n.split('::')[1].split('(')[0][:24]:24s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.2f} us")
for t in ("r03tree", "r05tree"):
    try:
        for r in csv.DictReader(open(f"gpurun_out/r05b_{t}_kernel_stats.csv")):
            if "k_masked_sgd_vec" in r["Name"] or "k_fullscan" in r["Name"] or "k_main" in r["Name"]:
                print(t, r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["StdDev"])
    except Exception as e:
        print(t, "ERR", e)
d = json.loads(open("gpurun_out/r05b_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["roofline"]["frac"], d["roofline"]["mean_launch_us"], d["roofline"]["event_pair_overhead_us"], d["mask_gen"])
PY
tail -c 800 gpurun_out/r05b_bench.err
