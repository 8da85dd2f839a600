cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_PORT=29635
run() {
  env "$@" timeout 600 python bench.py --gpus 1 --force_collectives --workload sd --steps 4 --warmup 2 --no_cpu_baseline > gpurun_out/sd_dp.json 2> gpurun_out/sd_dp.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/sd_dp.json') if l.startswith('{')][-1]); r=d.get('resident_activations') or {}; print('dp', '$*', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1), '| resident', round(r.get('ms_per_step'),2), round(r.get('host_enqueue_ms_per_step'),1))"
}
run A=1
run SALUN_DP_DIAG=nosidewait
run SALUN_SD_TARGET_OVERLAP=0
run TORCH_NCCL_AVOID_RECORD_STREAMS=1
run GPU_MAX_HW_QUEUES=16
