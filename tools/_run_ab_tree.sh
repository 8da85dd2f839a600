cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  ( cd build_lab/old && python bench.py --steps 177 --warmup 10 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])" )
  python bench.py --steps 177 --warmup 10 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
done
timeout 900 python -m pytest tests/test_ddpm_block_gpu.py tests/test_conv_gpu.py -q -m gpu --timeout 600 2>&1 | tail -2
timeout 600 python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ddpm', d['value'], d['ms_per_step'])"
