cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_bf16_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/convbench_bf16.py 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tools/bench_sd.py --bf16 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-330
