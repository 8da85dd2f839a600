mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_ring_gpu.py -x -q 2>&1 | tail -3
run() { # name env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3))"
}
NR=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so
for i in 1 2; do
run "wgrad_v  overlap prio0" SALUN_LIB=$NR
run "wgrad_v  overlap prio2" SALUN_LIB=$NR SALUN_RING_PRIO=2
run "wgrad_r  overlap prio0" X=1
run "wgrad_r  overlap prio1" SALUN_RING_PRIO=1
run "wgrad_r  overlap prio2" SALUN_RING_PRIO=2
run "wgrad_r  overlap prio3" SALUN_RING_PRIO=3
run "wgrad_r  serial  prio0" SALUN_WGRAD_OVERLAP=0
run "wgrad_v  serial  prio0" SALUN_LIB=$NR SALUN_WGRAD_OVERLAP=0
done
