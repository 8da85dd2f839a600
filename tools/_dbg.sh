cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/measure_tolerances.py tests/test_next_gpu.py tests/test_conv_gpu.py tests/test_norm_gpu.py tests/test_f4_gpu.py 2>&1 | grep -v amdgpu | tail -25
cat gpurun_out/r04_tolerance_use.txt
