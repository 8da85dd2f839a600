#!/bin/bash
# One rocprofv3 PMC pass (counters in their own run, kernel-trace only — never with sys/hip/hsa traces).
#   tools/pmc.sh <tag> <COUNTER> <command...>  -> gpurun_out/<tag>_<COUNTER>.csv : kernel, dispatches, mean counter
set -u
tag=$1; ctr=$2; shift 2
out=/tmp/pmc_${tag}_${ctr}
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
here=$(pwd)
args=()
for a in "$@"; do if [ -f "$here/$a" ]; then args+=("$here/$a"); else args+=("$a"); fi; done
( cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $tag -- "${args[@]}" ) > $out/run.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
mkdir -p $here/gpurun_out
if [ -z "$f" ]; then echo "no counter file"; tail -5 $out/run.log; exit 0; fi
python3 - "$f" "$ctr" "$here/gpurun_out/${tag}_${ctr}.csv" <<'PY'
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    k = r["Kernel_Name"][:90]
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
with open(sys.argv[3], "w") as f:
    f.write("kernel,dispatches,mean_%s\n" % sys.argv[2])
    for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%.3f\n' % (k, n, s / n))
print(open(sys.argv[3]).read()[:3000])
PY
