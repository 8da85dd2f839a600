cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_bf16_gpu.py tests/test_sd_parity_gpu.py tests/test_sd_gpu.py tests/test_tok_bf16_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python tools/bench_sd.py --bf16 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-300
SALUN_WGRAD_TN=0 timeout 600 python tools/bench_sd.py --bf16 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-300
