mkdir -p gpurun_out/r06
( time timeout 1500 python bench.py > gpurun_out/r06/bench_full.json 2> gpurun_out/r06/bench_full.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06/bench_full.json') if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], 'fwd_bwd', d['fwd_bwd']['frac'])
print('cpu', d.get('cpu_baseline',{}).get('value'))
for k in ('ddpm','sd'):
    b=d.get(k,{}); print(k, b.get('value'), b.get('ms_per_step'), b.get('error'), (b.get('cpu_baseline') or {}).get('value'), (b.get('cpu_baseline') or {}).get('ms_per_step_at_sampled_batch'))
print(json.dumps(d.get('dp_ws1'), indent=1)[:1500])
PY
tail -5 gpurun_out/r06/bench_full.err
