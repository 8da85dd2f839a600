cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_t && ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python $GRAFT_REPO_ROOT/tools/bench_sd.py --bf16 --steps 3 --warmup 2 > /dev/null 2>&1 )
t=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
idx=[i for i,r in enumerate(rows) if "k_masked_adam" in r["Kernel_Name"]]
print("adam launches", len(idx))
a,b=idx[-3],idx[-1]
seg=rows[a+1:b+1]; n=2
wall=(rows[b]["e"]-rows[a]["e"])/n/1e6
# union busy
iv=sorted((r["s"],r["e"]) for r in seg)
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print(f"steady steps: wall {wall:.1f} ms/step, union busy {busy/n/1e6:.1f} ms/step ({busy/n/1e6/wall:.3f}), kernels/step {len(seg)/n:.0f}")
q=collections.Counter(r.get("Queue_Id","") for r in seg); print("queues", q.most_common(4))
# gap histogram
gaps=[]; ce=iv[0][1]
for s,e in iv[1:]:
    if s>ce: gaps.append(s-ce)
    ce=max(ce,e)
gaps.sort(reverse=True)
print("idle total ms/step", sum(gaps)/n/1e6, "n gaps", len(gaps)/n, "top gaps us", [g/1e3 for g in gaps[:10]])
small=[g for g in gaps if g<20000]
print("gaps <20us: count/step", len(small)/n, "sum ms/step", sum(small)/n/1e6)
PY
