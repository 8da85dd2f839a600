# Round 5, GPU call 27: fp32 GEMM with an unguarded copy of its loop for interior tiles (loads really in flight during
# the multiply), ring epilogues (K16 GEMM / K11 ring forward) with the residual rows requested up front.  Parity suites
# of the touched kernels, then DDPM and SD against build_lab/base on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_bf16_gpu.py tests/test_conv_bf16_gpu.py tests/test_ddpm_block_gpu.py tests/test_ddpm_gpu.py tests/test_attn_gpu.py tests/test_sd_parity_gpu.py tests/test_sd_gpu.py tests/test_tok_bf16_gpu.py -x -q 2>&1 | tail -3 )
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],3), round(d['ms_per_step'],3))"; }
for rep in 1 2; do
  timeout 600 python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | tail -1 | one "ddpm   this"
  ( cd build_lab/base && timeout 600 python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | tail -1 | one "ddpm   base" )
  timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | one "sd     this"
  ( cd build_lab/base && timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | one "sd     base" )
done
timeout 300 python tools/gemmbench_f32.py 2>&1 | tail -25
