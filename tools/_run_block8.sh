cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ddpm_block_gpu.py tests/test_conv_gpu.py tests/test_ddpm_gpu.py tests/test_f4_gpu.py tests/test_sd_parity_gpu.py -q -m gpu --timeout 600 2>&1 | tail -12 > gpurun_out/blk8_tests.txt; cat gpurun_out/blk8_tests.txt
timeout 600 python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 20 --warmup 3 > gpurun_out/blk8_ddpm.json 2>gpurun_out/blk8_ddpm.err
python -c "
import json,sys; d=json.loads(open('gpurun_out/blk8_ddpm.json').read().strip().splitlines()[-1]); print('ddpm', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])" || tail -5 gpurun_out/blk8_ddpm.err
python bench.py --steps 60 --warmup 10 --no_cpu_baseline > gpurun_out/blk8_bench.json 2> gpurun_out/blk8_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/blk8_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
