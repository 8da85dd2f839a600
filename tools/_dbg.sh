cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_sd_parity_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/bench_sd.py --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-260
timeout 600 python tools/bench_ddpm.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-260
