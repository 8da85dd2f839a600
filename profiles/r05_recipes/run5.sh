# Round 5, GPU call 5: top-k parity + times after the conditional zero / min tracking and the deeper segment walk;
# smoke(); DDPM / SD mask paths (they rank one ratio at N_D / N_S through the same entry point).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_kernels_gpu.py -k "topk" -q 2>&1 | grep -v "amdgpu.ids" | tail -6 )
( timeout 900 python -m pytest tests/test_next_gpu.py -x -q 2>&1 | tail -4 )
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for cfg in "n18 1" "n18 10" "nd 1" "nd 10" "ns 1"; do
  set -- $cfg
  timeout 300 python tools/topk_prof.py $1 $2 20 2>&1 | grep "mask_topk n="
done
for cfg in "n18 1" "n18 10" "nd 1" "ns 1"; do
  set -- $cfg
  KEEP_TRACE=0 timeout 300 bash tools/prof.sh r05j_topk_$1_$2 python tools/topk_prof.py $1 $2 10 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r05j_*_kernel_stats.csv")):
    row = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "k_" in n and not any(s in n for s in ("fill", "popcount", "partials")):
            row[n.split("::")[1].split("(")[0]] = float(r["AverageNs"]) / 1e3
    print(f.split("r05j_")[1].split("_kernel")[0].ljust(14), "  ".join(f"{k[2:14]} {v:6.2f}" for k, v in sorted(row.items())), " sum %.1f" % sum(row.values()))
PY
