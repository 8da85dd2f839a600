# Round 6, part B: the whole -m gpu suite + smoke(), PMC traffic passes (counters in their own runs), SQ counter passes over
# the fp32 convolution kernels, the co-run / clock / data-parallel evidence.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r06_gpu_suite.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> gpurun_out/r06_gpu_suite.txt 2>&1
cat gpurun_out/r06_gpu_suite.txt
for sz in "n18 11173962" "nd 38632323" "ns 859520964"; do
  set -- $sz
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 400 bash tools/pmc.sh r06_$1 $ctr python tools/kbench_update.py $2 > /dev/null 2>&1
  done
done
timeout 300 bash tools/pmc_multi.sh r06_conv_sq_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python tools/pmc_conv_layers.py > /dev/null 2>&1
timeout 300 bash tools/pmc_multi.sh r06_conv_sq_b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" python tools/pmc_conv_layers.py > /dev/null 2>&1
( echo "# BatchNorm backward beside backward-weight (tools/corun_bench.py): ring kernel, then conv_wgrad_v (build_lab/noring)";
  timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids;
  SALUN_LIB=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids;
  echo "# sustained rates and the backward-data || backward-weight pair (tools/sustained_bench.py)";
  timeout 300 python tools/sustained_bench.py 2>&1 | grep -v "amdgpu.ids\|smi"
  echo "# shader clock under load (tools/clock_probe.py)";
  timeout 300 python tools/clock_probe.py 2>&1 | grep -v "amdgpu.ids\|Replacing\|random seed\|^45000" ) > gpurun_out/r06_corun.txt 2>&1
( for i in 1 2 3; do
  for cfg in "plain|" "dp|--force_collectives" "dp_no_probe_4q|--force_collectives"; do
    name=${cfg%%|*}; extra=${cfg##*|}
    envs=""; [ $name = dp_no_probe_4q ] && envs="GPU_MAX_HW_QUEUES=4 SALUN_STREAM_PROBE=0"
    env $envs timeout 200 python bench.py --steps 60 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen --no_sd --no_dp $extra 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet18 $name', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms')"
  done; done ) > gpurun_out/r06_dp_ws1.txt 2>&1
cat gpurun_out/r06_dp_ws1.txt gpurun_out/r06_corun.txt
python tools/pmc_traffic.py r06_n18:11173962 r06_nd:38632323 r06_ns:859520964 > gpurun_out/r06_pmc_traffic.json 2>/dev/null; head -c 600 gpurun_out/r06_pmc_traffic.json
