cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_target_overlap_gpu.py tests/test_sd_parity_gpu.py tests/test_sd_gpu.py -q -m gpu --timeout 600 2>&1 | tail -15 > gpurun_out/blk4_tests.txt; cat gpurun_out/blk4_tests.txt
for ov in 0 1; do
  SALUN_SD_TARGET_OVERLAP=$ov timeout 900 python tools/bench_sd.py --bf16 --steps 4 --warmup 2 > gpurun_out/blk4_sd_$ov.json 2>gpurun_out/blk4_sd_$ov.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/blk4_sd_$ov.json').read().strip().splitlines()[-1]); print('sd overlap=$ov', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/blk4_sd_$ov.err
done
