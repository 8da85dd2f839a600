run() { name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3))"
}
NR=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so
for i in 1 2; do
run "wgrad_v" SALUN_LIB=$NR
run "wgrad_r prio0" X=1
run "wgrad_r prio1" SALUN_WGR_PRIO=1
run "wgrad_r prio2" SALUN_WGR_PRIO=2
run "wgrad_r prio3" SALUN_WGR_PRIO=3
done
echo corun; SALUN_WGR_PRIO=3 timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids
