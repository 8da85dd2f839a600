# Round 5, GPU call 21: grid cap of conv_wgrad_reduce (grid-stride form) on the ResNet-18 and DDPM steps, one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -2 )
for c in 0 512 1024 2048 0 1024; do
  SALUN_WGRAD_REDUCE_GRID=$c timeout 300 python bench.py --steps 177 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet cap $c', round(d['value'],2), round(d['ms_per_step'],3))"
done
for c in 0 1024 0 1024; do
  SALUN_WGRAD_REDUCE_GRID=$c timeout 300 python tools/bench_ddpm.py --steps 10 --warmup 3 --mask_batches 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ddpm cap $c', round(d['value'],3), round(d['ms_per_step'],2))"
done
