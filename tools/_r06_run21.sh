cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in plain dp; do
  extra=""; [ $mode = dp ] && extra="--force_collectives"
  KEEP_TRACE=1 timeout 400 bash tools/prof.sh r06c_ddpm_$mode python bench.py --workload ddpm --steps 4 --warmup 2 --no_cpu_baseline --no_mask_gen --ddpm_mask_batches 2 $extra > /dev/null 2>&1
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r06c_ddpm_${mode}_trace_slim.csv')))
for r in rows: r['s']=int(r['start_ns']); r['e']=int(r['end_ns'])
idx=[i for i,r in enumerate(rows) if 'k_masked_adam' in r['name']]
a,b=idx[-3],idx[-2]
seg=rows[a+1:b+1]
wall=(rows[b]['e']-rows[a]['e'])/1e3
iv=sorted((r['s'],r['e']) for r in seg)
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
tot=sum(r['e']-r['s'] for r in seg)/1e3
print('$mode', 'wall', round(wall), 'us kernels', len(seg), 'union busy', round(busy/1e3), 'sum of durations', round(tot), 'overlap factor', round(tot/(busy/1e3),3))
fam=collections.defaultdict(float)
for r in seg:
    n=r['name'].replace('void ','').replace('(anonymous namespace)::','').split('<')[0].split('(')[0][:32]
    fam[n]+=(r['e']-r['s'])/1e3
print(sorted(((round(v),k) for k,v in fam.items()), reverse=True)[:10])
PY
done
