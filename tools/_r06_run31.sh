timeout 1500 python -m pytest tests/test_rccl_ws1_gpu.py tests/test_dist_gpu.py tests/test_dist_diffusion_gpu.py tests/test_target_overlap_gpu.py -x -q 2>&1 | tail -3
run() { name=$1; shift
  timeout 200 python bench.py --steps 60 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen --no_sd --no_dp "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3))"
}
for i in 1 2 3; do run plain; run dp --force_collectives; done
timeout 300 python bench.py --workload ddpm --steps 8 --warmup 3 --no_cpu_baseline --no_mask_gen --ddpm_mask_batches 2 --force_collectives 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ddpm dp', round(d['value'],3), round(d['ms_per_step'],2))"
