#!/bin/bash
# One rocprofv3 PMC pass with several counters of one block group (own run, kernel-trace only).
#   tools/pmc_multi.sh <tag> "<C1 C2 ...>" <command...>  -> gpurun_out/<tag>_pmc.csv : kernel, dispatches, mean of each counter
set -u
tag=$1; ctrs=$2; shift 2
out=/tmp/pmcm_${tag}
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
here=$(pwd)
args=()
for a in "$@"; do if [ -f "$here/$a" ]; then args+=("$here/$a"); else args+=("$a"); fi; done
( cd /tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o $tag -- "${args[@]}" ) > $out/run.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
mkdir -p $here/gpurun_out
if [ -z "$f" ]; then echo "no counter file"; tail -5 $out/run.log; exit 0; fi
python3 - "$f" "$here/gpurun_out/${tag}_pmc.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]; c = r["Counter_Name"]
    if c not in names: names.append(c)
    acc[k][c][0] += 1; acc[k][c][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k, d in acc.items():
        n = max(v[0] for v in d.values())
        f.write('"%s",%d,' % (k, n) + ",".join("%.4g" % (d[c][1] / max(d[c][0], 1)) for c in names) + "\n")
print(open(sys.argv[2]).read()[:4000])
PY
