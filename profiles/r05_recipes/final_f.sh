# Round 5, the last two GPU-minutes: the SQ counters of the 3x3 SD layer again, now with the paired backward-weight
# kernel and the batched epilogues (same recipe as final_c.sh; the forward / backward-data rows should not move).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
SALUN_CONV_RING=0 timeout 50 bash tools/pmc_multi.sh r05_conv3x3_final_a "$A" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
SALUN_CONV_RING=0 timeout 50 bash tools/pmc_multi.sh r05_conv3x3_final_b "$B" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
grep -h "igemm<\|conv_bf16_wgrad\|^kernel" gpurun_out/r05_conv3x3_final_a_pmc.csv gpurun_out/r05_conv3x3_final_b_pmc.csv | cut -c1-230
