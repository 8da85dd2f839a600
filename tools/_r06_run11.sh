timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_ring_gpu.py tests/test_fullsize_gpu.py tests/test_classification_gpu.py tests/test_ddpm_block_gpu.py tests/test_dist_gpu.py tests/test_rccl_ws1_gpu.py -x -q 2>&1 | tail -4
run() { name=$1; shift
  timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3))"
}
for i in 1 2; do
run "plain"
run "force_collectives" --force_collectives
done
