cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_bf16_gpu.py -x -q -m gpu -k "tn or conv_1x1" 2>&1 | tail -2
bash tools/pmc.sh r04_new FETCH_SIZE python tools/gemm_pmc.py > /dev/null 2>&1
bash tools/pmc.sh r04_new WRITE_SIZE python tools/gemm_pmc.py > /dev/null 2>&1
python tools/pmc_traffic.py r04_new:k_gemm_bf16_tn=196935680 | grep -E "traffic_bytes|over"
timeout 600 python tools/gemmbench_bf16.py --reps 20 2>&1 | grep -v amdgpu > gpurun_out/r04_gemmbench_bf16.txt; tail -17 gpurun_out/r04_gemmbench_bf16.txt
