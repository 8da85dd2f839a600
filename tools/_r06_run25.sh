cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_ring_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | tail -2
timeout 300 bash tools/pmc_multi.sh r06_conv_sq_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python tools/pmc_conv_layers.py > /dev/null 2>&1
grep "conv3x3\|kernel," gpurun_out/r06_conv_sq_a_pmc.csv | cut -c1-200
timeout 300 python tools/convring_bench.py --cfgs 0 2>&1 | grep "ring cfg\|igemm"
timeout 200 python tools/wgradbench.py 2>&1 | grep -v amdgpu
