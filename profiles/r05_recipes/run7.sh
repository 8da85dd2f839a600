cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for e in 0 32; do for sz in ns nd n18; do
  SALUN_TOPK_EXP=$e KEEP_TRACE=0 timeout 200 bash tools/prof.sh r05g_${sz}_exp${e} python tools/topk_prof.py $sz 1 10 > /dev/null 2>&1
done; done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r05g_*_kernel_stats.csv")):
    row = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "k_" in n and not any(s in n for s in ("fill", "popcount", "partials")):
            row[n.split("::")[1].split("(")[0]] = float(r["AverageNs"]) / 1e3
    print(f.split("r05g_")[1].split("_kernel")[0].ljust(14), "  ".join(f"{k[2:14]} {v:6.2f}" for k, v in sorted(row.items())), " sum %.1f" % sum(row.values()))
PY
