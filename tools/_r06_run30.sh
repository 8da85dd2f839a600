run() { name=$1; shift
  env "$@" timeout 200 python bench.py --steps 60 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen --no_sd --no_dp 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
run "product" X=1
run "no folds (upper bound)" SALUN_LIB=$PWD/build_lab/nofold/unlearn_saliency_amd/libsalun.so
done
