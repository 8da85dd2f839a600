P=$PWD/build_lab/prev/unlearn_saliency_amd/libsalun.so
NR=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so
timeout 600 python -m pytest tests/test_norm_gpu.py -x -q 2>&1 | tail -2
echo "== corun prev (BN with divisions), ring wgrad (not shared flag => ring)"; SALUN_LIB=$P timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids
echo "== corun new"; timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen --no_sd --no_dp 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
run "prev" SALUN_LIB=$P
run "new BN" X=1
done
