cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ddpm_gpu.py tests/test_sd_parity_gpu.py tests/test_f4_gpu.py tests/test_fullsize_diffusion_gpu.py -q -m gpu --timeout 900 -s 2>&1 | grep -v "it/s\|s/it\|^Replacing\|^setup random\|Loading checkpoints" > gpurun_out/r3g_tests.txt
grep -n "max |err|\|of scale\|differ\|moved\|rel \[\|passed\|failed\|Error\|error\|FAILED\|positions\|reference run" gpurun_out/r3g_tests.txt | cut -c1-260 | tail -60
