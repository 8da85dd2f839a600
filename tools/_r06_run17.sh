run() { name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen $EXTRA 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3), d['roofline']['mean_launch_us'])"
}
for i in 1 2 3 4 5; do
EXTRA="--force_collectives"; run "dp (8 queues)" X=1
EXTRA="--force_collectives"; run "dp (4 queues)" GPU_MAX_HW_QUEUES=4
done
EXTRA=""; run "plain" X=1
timeout 600 python -m pytest tests/test_rccl_ws1_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | tail -3
