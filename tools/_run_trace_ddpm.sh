cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_t && ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python $GRAFT_REPO_ROOT/tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 6 --warmup 3 > /dev/null 2>&1 )
t=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python3 - "$t" gpurun_out/r3_trace_ddpm_slim.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-9000:]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_ns,end_ns,queue,name\n")
    for r in rows:
        f.write(f'{int(r["Start_Timestamp"])-t0},{int(r["End_Timestamp"])-t0},{r.get("Queue_Id","")},"{r["Kernel_Name"][:60]}"\n')
PY
python3 tools/trace_summary.py gpurun_out/r3_trace_ddpm_slim.csv
