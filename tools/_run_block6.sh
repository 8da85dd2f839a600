cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 20 --warmup 3 > gpurun_out/blk6_ddpm.json 2>gpurun_out/blk6_ddpm.err
python -c "
import json,sys; d=json.loads(open('gpurun_out/blk6_ddpm.json').read().strip().splitlines()[-1]); print('ddpm', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])" || tail -5 gpurun_out/blk6_ddpm.err
SALUN_SYNC_DEBUG=1 timeout 600 python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 1 --steps 2 --warmup 2 2>&1 >/dev/null | grep -i "synchroniz" | sort | uniq -c | head
timeout 600 python -m pytest tests/test_ddpm_gpu.py tests/test_target_overlap_gpu.py -q -m gpu -x 2>&1 | tail -2
