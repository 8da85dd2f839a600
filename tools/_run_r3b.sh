set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sd_parity_gpu.py -x -q -m gpu --timeout 600 -s 2>&1 | grep -v "it/s\|s/it" | tail -40 > gpurun_out/r3b_tests.txt
tail -25 gpurun_out/r3b_tests.txt
timeout 300 python tools/convbench.py --no_lib > gpurun_out/r3b_convbench.txt 2>&1; cat gpurun_out/r3b_convbench.txt
bash tools/pmc_multi.sh r3b_conv "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" python tools/convbench.py --no_lib | cut -c1-400
bash tools/pmc_multi.sh r3b_conv2 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" python tools/convbench.py --no_lib | cut -c1-400
