run() { name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen --force_collectives 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3), d['roofline']['mean_launch_us'])"
}
run "dp default" X=1
run "dp hwq8" GPU_MAX_HW_QUEUES=8
run "dp hwq16" GPU_MAX_HW_QUEUES=16
run "dp hwq2" GPU_MAX_HW_QUEUES=2
timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', round(d['value'],2), round(d['ms_per_step'],3))"
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain hwq8', round(d['value'],2), round(d['ms_per_step'],3))"
