cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ddpm_gpu.py tests/test_fullsize_diffusion_gpu.py -q -m gpu -x --timeout 600 -k "not sd_v1" 2>&1 | tail -5 > gpurun_out/blk3_tests.txt; cat gpurun_out/blk3_tests.txt
for ov in 0 1; do
  SALUN_DDPM_TARGET_OVERLAP=$ov timeout 600 python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 20 --warmup 3 > gpurun_out/blk3_ddpm_$ov.json 2>gpurun_out/blk3_ddpm_$ov.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/blk3_ddpm_$ov.json').read().strip().splitlines()[-1]); print('ddpm overlap=$ov', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/blk3_ddpm_$ov.err
done
