# Round-4 measurements: benches, rocprofv3 kernel stats, counter passes.  Everything lands in gpurun_out/r04_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python bench.py --forget class --no_cpu_baseline > gpurun_out/r04_bench_class.json 2>/dev/null
python bench.py --workload ddpm > gpurun_out/r04_ddpm_bench.json 2>/dev/null
python bench.py --workload sd --steps 6 --warmup 2 > gpurun_out/r04_sd_bench_bf16.json 2>/dev/null
python tools/bench_sd.py --steps 3 --warmup 1 > gpurun_out/r04_sd_bench_f32.json 2>/dev/null
prof() {  # tag, command...
  tag=$1; shift
  rm -rf /tmp/prof_$tag && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" > /dev/null 2>&1 )
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04_${tag}_kernel_stats.csv
}
prof bench python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --steps 177
prof ddpm python $GRAFT_REPO_ROOT/tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 10 --warmup 3
prof sd_bf16 python $GRAFT_REPO_ROOT/tools/bench_sd.py --bf16 --steps 5 --warmup 2
python tools/gemmbench_bf16.py --reps 20 2>&1 | grep -v amdgpu > gpurun_out/r04_gemmbench_bf16.txt
python tools/convbench.py --no_lib 2>&1 | grep -v amdgpu > gpurun_out/r04_convbench.txt
python tools/convbench_bf16.py 2>&1 | grep -v amdgpu > gpurun_out/r04_convbench_bf16.txt
python tools/kbench.py --sizes n18,nd,ns --iters 30 --extra --json gpurun_out/r04_kbench.json > gpurun_out/r04_kbench.txt 2>&1
# counters (own runs, kernel-trace only)
bash tools/pmc_multi.sh r04_gemm_sq_a "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" python tools/gemm_pmc.py > /dev/null 2>&1
bash tools/pmc_multi.sh r04_gemm_sq_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" python tools/gemm_pmc.py > /dev/null 2>&1
for sz in "n18 11173962" "nd 38632323"; do set -- $sz
  bash tools/pmc.sh r04_$1 FETCH_SIZE python tools/kbench_update.py $2 > /dev/null 2>&1
  bash tools/pmc.sh r04_$1 WRITE_SIZE python tools/kbench_update.py $2 > /dev/null 2>&1
done
bash tools/pmc.sh r04_ns FETCH_SIZE python tools/topk_prof.py ns 1 3 > /dev/null 2>&1
bash tools/pmc.sh r04_ns WRITE_SIZE python tools/topk_prof.py ns 1 3 > /dev/null 2>&1
bash tools/pmc.sh r04_new FETCH_SIZE python tools/gemm_pmc.py > /dev/null 2>&1
bash tools/pmc.sh r04_new WRITE_SIZE python tools/gemm_pmc.py > /dev/null 2>&1
# algorithmic bytes of one launch of tools/gemm_pmc.py's shapes: NT (M K + N K + M N) 2; TN (M Na + M Nb) 2 + 2 Na Nb 4;
# K15 (M K + N K + M N) 4; dropout 8 n
python tools/pmc_traffic.py r04_n18:11173962 r04_nd:38632323 r04_ns:859520964 \
  r04_new:k_gemm_bf16_nt_r=190382080 r04_new:k_gemm_bf16_tn=196935680 r04_new:k_gemm_f32=37748736 r04_new:k_dropout=134217728 \
  > gpurun_out/r04_pmc_traffic.json
python - <<'PY'
import json
for f in ("r04_bench","r04_bench_class","r04_ddpm_bench","r04_sd_bench_bf16","r04_sd_bench_f32"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],3), round(d["ms_per_step"],3), d.get("roofline",{}).get("frac"), (d.get("fwd_bwd") or {}).get("frac", (d.get("fwd_bwd") or {}).get("frac_whole_step")))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r04_gemmbench_bf16.txt; head -c 600 gpurun_out/r04_pmc_traffic.json
