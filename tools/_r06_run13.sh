timeout 1500 python -m pytest tests/test_rccl_ws1_gpu.py tests/test_dist_gpu.py tests/test_dist_diffusion_gpu.py -x -q 2>&1 | tail -4
run() { name=$1; shift
  timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3), d['roofline']['mean_launch_us'])"
}
for i in 1 2; do
run "plain"
run "force_collectives" --force_collectives
done
