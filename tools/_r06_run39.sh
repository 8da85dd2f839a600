cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_PORT=29631
for ov in 1 0; do
SALUN_BF16_WGRAD_OVERLAP=$ov timeout 600 python bench.py --gpus 1 --force_collectives --workload sd --steps 4 --warmup 2 --no_cpu_baseline > gpurun_out/sd_dp.json 2> gpurun_out/sd_dp.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/sd_dp.json') if l.startswith('{')][-1]); r=d.get('resident_activations') or {}; print('dp overlap=$ov', round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1), d.get('collectives'), '| resident', r.get('ms_per_step'), r.get('host_enqueue_ms_per_step'))"
done
timeout 600 python bench.py --gpus 1 --workload sd --steps 4 --warmup 2 --no_cpu_baseline > gpurun_out/sd_dp.json 2> gpurun_out/sd_dp.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/sd_dp.json') if l.startswith('{')][-1]); r=d.get('resident_activations') or {}; print('plain', round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1), d.get('collectives'), '| resident', r.get('ms_per_step'), r.get('host_enqueue_ms_per_step'))"
