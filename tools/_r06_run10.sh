NR=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so
run() { name=$1; shift
  env "$@" timeout 300 python tools/bench_ddpm.py --steps 8 --warmup 3 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],3), round(d['ms_per_step'],2))"
}
for i in 1 2; do
run "ddpm igemm+wgrad_v " SALUN_LIB=$NR SALUN_RING=0
run "ddpm ring +wgrad_v " SALUN_LIB=$NR
run "ddpm ring +wgrad_r " X=1
run "ddpm ring +wgrad_r serial" SALUN_WGRAD_OVERLAP=0
done
