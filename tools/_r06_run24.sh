cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 bash tools/pmc_multi.sh r06_conv_sq_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python tools/pmc_conv_layers.py > /dev/null 2>&1
timeout 300 bash tools/pmc_multi.sh r06_conv_sq_b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" python tools/pmc_conv_layers.py > /dev/null 2>&1
grep "conv3x3\|conv_igemm\|conv_wgrad_v\|kernel," gpurun_out/r06_conv_sq_a_pmc.csv | cut -c1-260
grep "conv3x3\|conv_igemm\|conv_wgrad_v\|kernel," gpurun_out/r06_conv_sq_b_pmc.csv | cut -c1-260
