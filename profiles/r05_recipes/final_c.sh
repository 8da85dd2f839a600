# Round 5, last GPU call: the whole -m gpu suite and smoke() on the final tree, the default bench line, the SD bf16 line,
# and the SQ counters of one 3x3 SD layer on the descriptor-staged K11 kernels (same recipe as r04_conv3x3_staged_*).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 ) > gpurun_out/r05_gpu_suite.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> gpurun_out/r05_gpu_suite.txt 2>&1
cat gpurun_out/r05_gpu_suite.txt
( time timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err ) 2>&1 | tail -3
timeout 900 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline > gpurun_out/r05_sd_bench_bf16.json 2>/dev/null
A="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
SALUN_CONV_RING=0 bash tools/pmc_multi.sh r05_conv3x3_a "$A" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
SALUN_CONV_RING=0 bash tools/pmc_multi.sh r05_conv3x3_b "$B" python tools/convlayer_bf16.py 32 640 640 3 1 > /dev/null 2>&1
grep -h "igemm<\|conv_bf16_wgrad\|^kernel" gpurun_out/r05_conv3x3_a_pmc.csv gpurun_out/r05_conv3x3_b_pmc.csv | cut -c1-230
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_net_of_event_overhead"], d["fwd_bwd"]["frac"])
print("mask_gen", {k: v for k, v in d["mask_gen"].items() if k != "topk_note"})
print("ddpm", {k: v for k, v in d["ddpm"].items() if k in ("value", "ms_per_step", "error")})
s = json.loads(open("gpurun_out/r05_sd_bench_bf16.json").read().strip().splitlines()[-1])
print("sd", s["value"], s["ms_per_step"], s["roofline"]["traffic"])
PY
