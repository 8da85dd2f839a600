# SQ counters of one 3x3 SD layer (8 x 32 x 32, 640 -> 640) on the register-staged forward kernel and, with
# SALUN_CONV_RING=3, on the LDS-DMA ring kernel (the A/B behind "the ring loses the 3x3 layers")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
SALUN_CONV_RING=0 bash tools/pmc_multi.sh r04_conv3x3_staged_a "$A" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
SALUN_CONV_RING=3 bash tools/pmc_multi.sh r04_conv3x3_ring_a "$A" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
SALUN_CONV_RING=0 bash tools/pmc_multi.sh r04_conv3x3_staged_b "$B" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
SALUN_CONV_RING=3 bash tools/pmc_multi.sh r04_conv3x3_ring_b "$B" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
grep -h "igemm<\|conv_bf16_ring\|^kernel" gpurun_out/r04_conv3x3_staged_a_pmc.csv gpurun_out/r04_conv3x3_ring_a_pmc.csv gpurun_out/r04_conv3x3_staged_b_pmc.csv gpurun_out/r04_conv3x3_ring_b_pmc.csv | cut -c1-230
