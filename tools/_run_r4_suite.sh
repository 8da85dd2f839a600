cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r04_measured_errors.txt
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -25 > gpurun_out/r04_gpu_suite.txt; cat gpurun_out/r04_gpu_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
