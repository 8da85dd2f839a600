# Round 5, GPU call 29: fp32 backward-weight reduce writing whole OIHW rows (3x3, >= 32768 (k, c) pairs) against
# build_lab/norows (same tree, -DSALUN_WGRAD_REDUCE_ROWS=0): parity of the fp32 convolutions, then the ResNet-18 and
# DDPM steps alternated on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_classification_gpu.py tests/test_ddpm_block_gpu.py -x -q 2>&1 | tail -2 )
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],3), round(d['ms_per_step'],3))"; }
for rep in 1 2; do
  timeout 600 python bench.py --no_cpu_baseline --no_ddpm 2>/dev/null | tail -1 | one "resnet rows  "
  ( cd build_lab/norows && timeout 600 python bench.py --no_cpu_baseline --no_ddpm 2>/dev/null | tail -1 | one "resnet norows" )
done
timeout 600 python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | tail -1 | one "ddpm   rows  "
( cd build_lab/norows && timeout 600 python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | tail -1 | one "ddpm   norows" )
