#!/bin/bash
# rocprofv3 kernel-trace of a command on the GPU box; keeps only the small CSV summaries.
#   tools/prof.sh <tag> <command...>      -> gpurun_out/<tag>_kernel_stats.csv (+ top of the table on stdout)
set -u
tag=$1; shift
out=/tmp/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
here=$(pwd)
args=()
for a in "$@"; do if [ -f "$here/$a" ]; then args+=("$here/$a"); else args+=("$a"); fi; done
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- "${args[@]}" ) > $out/run.log 2>&1
mkdir -p $here/gpurun_out
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $here/gpurun_out/${tag}_kernel_stats.csv; head -25 $f | cut -c1-200; else echo "no stats file"; tail -20 $out/run.log; fi
t=$(find $out -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ] && [ "${KEEP_TRACE:-0}" = "1" ]; then python3 - "$t" "$here/gpurun_out/${tag}_trace_slim.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    f.write("start_ns,end_ns,name\n")
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows[-6000:]:
        f.write(f'{int(r["Start_Timestamp"])-t0},{int(r["End_Timestamp"])-t0},"{r["Kernel_Name"][:60]}"\n')
PY
fi
tail -3 $out/run.log | cut -c1-300
