# Round 5, GPU call 26: epilogues that read their terms (bias / per-image bias / addend / old gradient) BEFORE the first
# store instead of one load-wait-store per element — fp32 conv_igemm / conv_igemm_tap / conv_dgrad_s2, fp32 and bf16 TN
# GEMM `+=`, bf16 conv_bf16_igemm, bf16 backward-weight `+=`.  Parity suites of every touched kernel, then the three
# workloads against build_lab/base (the tree before these changes) on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_gemm_gpu.py tests/test_gemm_bf16_gpu.py tests/test_conv_bf16_gpu.py tests/test_ddpm_block_gpu.py tests/test_ddpm_gpu.py tests/test_classification_gpu.py tests/test_f4_gpu.py -x -q 2>&1 | tail -3 )
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],3), round(d['ms_per_step'],3))"; }
for rep in 1 2; do
  timeout 600 python bench.py --no_cpu_baseline --no_ddpm 2>/dev/null | tail -1 | one "resnet this"
  ( cd build_lab/base && timeout 600 python bench.py --no_cpu_baseline --no_ddpm 2>/dev/null | tail -1 | one "resnet base" )
  timeout 600 python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | tail -1 | one "ddpm   this"
  ( cd build_lab/base && timeout 600 python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | tail -1 | one "ddpm   base" )
done
timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | one "sd     this"
( cd build_lab/base && timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | one "sd     base" )
timeout 300 python tools/convbench_bf16.py --iters 20 2>&1 | grep "total\|8x8\|16x16 1280->1280  3x3 s1"
