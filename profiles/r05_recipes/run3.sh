# Round 5, GPU call 3: where the time of k_main / k_resolve / k_finish goes (SALUN_TOPK_EXP switches cut the kernels
# short; results are wrong under a switch — timing only), grid 512 vs 1024, and the concentration tests again.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -k "topk_local or topk_moderate or topk_zero" -q 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r05_run3_tests.txt 2>&1
cat gpurun_out/r05_run3_tests.txt
for e in 0 1 2 4 8 16 24 64 128; do
  SALUN_TOPK_EXP=$e KEEP_TRACE=0 timeout 200 bash tools/prof.sh r05c_exp${e} python tools/topk_prof.py n18 1 10 > /dev/null 2>&1
done
SALUN_TOPK_GRID=512 KEEP_TRACE=0 timeout 200 bash tools/prof.sh r05c_grid512 python tools/topk_prof.py n18 1 10 > /dev/null 2>&1
for e in 0 8 16; do
  SALUN_TOPK_EXP=$e KEEP_TRACE=0 timeout 200 bash tools/prof.sh r05c_nk10_exp${e} python tools/topk_prof.py n18 10 10 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r05c_*_kernel_stats.csv")):
    row = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "k_" in n and not any(s in n for s in ("fill", "popcount", "partials")):
            row[n.split("::")[1].split("(")[0].split("<")[0]] = float(r["AverageNs"]) / 1e3
    print(f.split("r05c_")[1].split("_kernel")[0].ljust(12), "  ".join(f"{k[2:]} {v:6.2f}" for k, v in sorted(row.items())))
PY
