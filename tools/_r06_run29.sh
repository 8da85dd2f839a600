for v in 0 1 3; do
  L=$PWD/unlearn_saliency_amd/libsalun.so; [ $v != 0 ] && L=$PWD/build_lab/bnprio$v/unlearn_saliency_amd/libsalun.so
  echo "== BN prio $v"; SALUN_LIB=$L timeout 300 python tools/corun_bench.py 2>&1 | grep "co-run"
done
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --steps 60 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen --no_sd --no_dp 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
run "prio0" X=1
run "prio1" SALUN_LIB=$PWD/build_lab/bnprio1/unlearn_saliency_amd/libsalun.so
run "prio3" SALUN_LIB=$PWD/build_lab/bnprio3/unlearn_saliency_amd/libsalun.so
done
