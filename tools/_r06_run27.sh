run() { name=$1; shift
  env "$@" timeout 300 python bench.py --workload ddpm --steps 8 --warmup 3 --no_cpu_baseline --no_mask_gen --ddpm_mask_batches 2 $EXTRA 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],3), round(d['ms_per_step'],2), 'host', round(d.get('host_enqueue_ms_per_step',0),1))"
}
EXTRA=""; run "plain" X=1
EXTRA="--force_collectives"
run "dp" X=1
run "dp no target overlap" SALUN_DDPM_TARGET_OVERLAP=0
run "dp hwq16" GPU_MAX_HW_QUEUES=16
run "dp no wgrad overlap" SALUN_WGRAD_OVERLAP=0
run "dp no probe" SALUN_STREAM_PROBE=0
