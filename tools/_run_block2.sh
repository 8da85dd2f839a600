cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_ddpm_block_gpu.py tests/test_ddpm_gpu.py -q -m gpu -x --timeout 600 2>&1 | tail -8 > gpurun_out/blk2_tests.txt; cat gpurun_out/blk2_tests.txt
timeout 300 python tools/kbench.py --sizes n18 --extra 2>&1 | grep -E "gn_|bn_" > gpurun_out/blk2_kbench.txt; cat gpurun_out/blk2_kbench.txt
timeout 600 python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 20 --warmup 3 > gpurun_out/blk2_ddpm.json 2>gpurun_out/blk2_ddpm.err
python -c "
import json,sys; d=json.loads(open('gpurun_out/blk2_ddpm.json').read().strip().splitlines()[-1]); print('ddpm', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/blk2_ddpm.err
