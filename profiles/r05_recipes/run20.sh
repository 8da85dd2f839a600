# Round 5, GPU call 20: the SD bf16 step with the descriptor-staged K11 kernels — parity suites of the SD path and the
# step time of this tree vs build_lab/base (HEAD before the change) on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_sd_parity_gpu.py tests/test_sd_gpu.py tests/test_conv_bf16_gpu.py tests/test_fullsize_diffusion_gpu.py -x -q 2>&1 | tail -3 )
for rep in 1 2; do
  timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree', round(d['value'],3), round(d['ms_per_step'],2))"
  ( cd build_lab/base && timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base     ', round(d['value'],3), round(d['ms_per_step'],2))" )
done
