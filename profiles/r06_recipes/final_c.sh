# Round 6, last pass on the final tree: part A again (bench lines, kernel tables, layer tables), the -m gpu suite + smoke,
# and the co-run table with the conv_wgrad_v build (build_lab/noring = this tree with -DSALUN_WGRAD_NO_RING=1).
bash profiles/r06_recipes/final_a.sh > gpurun_out/r06_final_a.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 ) > gpurun_out/r06_gpu_suite.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> gpurun_out/r06_gpu_suite.txt 2>&1
( echo "# BatchNorm backward beside backward-weight (tools/corun_bench.py): ring kernel, then conv_wgrad_v (build_lab/noring)";
  timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids;
  SALUN_LIB=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids;
  echo "# sustained rates and the backward-data || backward-weight pair (tools/sustained_bench.py): ring backward-weight, then conv_wgrad_v";
  timeout 300 python tools/sustained_bench.py 2>&1 | grep -v "amdgpu.ids\|smi\|GPU\["
  SALUN_LIB=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so timeout 300 python tools/sustained_bench.py 2>&1 | grep -v "amdgpu.ids\|smi\|GPU\["
  echo "# shader clock under load (tools/clock_probe.py)";
  timeout 300 python tools/clock_probe.py 2>&1 | grep -v "amdgpu.ids\|Replacing\|random seed\|^45000" ) > gpurun_out/r06_corun.txt 2>&1
tail -25 gpurun_out/r06_final_a.log; cat gpurun_out/r06_gpu_suite.txt gpurun_out/r06_corun.txt
