cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_bf16_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/gemmbench_bf16.py --reps 20 > gpurun_out/r4e_gemmbench.txt 2>&1
timeout 600 python tools/bench_sd.py --bf16 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-300
