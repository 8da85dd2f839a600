# Round 5, GPU call 4: all top-k tests after the one-pass compaction / segment walk, per-kernel times at every size,
# k_resolve workgroup-count sweep.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_kernels_gpu.py -k "topk" -q 2>&1 | grep -v "amdgpu.ids" | tail -12 ) > gpurun_out/r05_run4_tests.txt 2>&1
( timeout 600 python -m pytest tests/test_next_gpu.py -x -q 2>&1 | tail -3 ) >> gpurun_out/r05_run4_tests.txt 2>&1
cat gpurun_out/r05_run4_tests.txt
for cfg in "n18 1" "n18 10" "nd 1" "nd 10" "ns 1"; do
  set -- $cfg
  timeout 300 python tools/topk_prof.py $1 $2 20 2>&1 | grep "mask_topk n="
done
for cfg in "n18 1" "n18 10" "nd 1" "nd 10" "ns 1"; do
  set -- $cfg
  KEEP_TRACE=0 timeout 300 bash tools/prof.sh r05d_topk_$1_$2 python tools/topk_prof.py $1 $2 10 > /dev/null 2>&1
done
for g in 16 32 64; do
  SALUN_TOPK_GR=$g KEEP_TRACE=0 timeout 200 bash tools/prof.sh r05d_gr${g} python tools/topk_prof.py n18 1 10 > /dev/null 2>&1
  SALUN_TOPK_GR=$g KEEP_TRACE=0 timeout 200 bash tools/prof.sh r05d_nd_gr${g} python tools/topk_prof.py nd 1 10 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r05d_*_kernel_stats.csv")):
    row = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "k_" in n and not any(s in n for s in ("fill", "popcount", "partials")):
            row[n.split("::")[1].split("(")[0]] = float(r["AverageNs"]) / 1e3
    print(f.split("r05d_")[1].split("_kernel")[0].ljust(14), "  ".join(f"{k[2:14]} {v:6.2f}" for k, v in sorted(row.items())), " sum %.1f" % sum(row.values()))
PY
timeout 600 python bench.py --no_cpu_baseline --no_ddpm 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['mask_gen'])"
