# Round 5, GPU call 10: folds of backward-weight's partial sums on a third stream — parity tests and the step A/B.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_classification_gpu.py tests/test_ddpm_block_gpu.py -x -q 2>&1 | tail -4 )
for f in 1 0 1 0; do
  SALUN_WGRAD_FOLD_STREAM=$f timeout 300 python bench.py --steps 177 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fold_stream $f', round(d['value'],2), round(d['ms_per_step'],3), round(d['fwd_bwd']['frac'],4), round(d['roofline']['mean_launch_us'],1))"
done
for f in 1 0; do
  SALUN_WGRAD_FOLD_STREAM=$f timeout 300 python tools/bench_ddpm.py --steps 10 --warmup 3 --mask_batches 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ddpm fold_stream $f', round(d['value'],3), round(d['ms_per_step'],2))"
done
