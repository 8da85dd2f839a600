cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -6 > gpurun_out/final_suite.txt; cat gpurun_out/final_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python bench.py --forget class --no_cpu_baseline > gpurun_out/final_bench_class.json 2>/dev/null
python bench.py --workload ddpm --no_cpu_baseline > gpurun_out/final_ddpm.json 2>/dev/null
python bench.py --workload sd --steps 5 --warmup 2 > gpurun_out/final_sd.json 2>/dev/null
rm -rf /tmp/prof_b && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --steps 177 > /dev/null 2>&1 )
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/final_bench_kernel_stats.csv
python tools/convbench.py --no_lib > gpurun_out/final_convbench.txt 2>&1
rm -rf /tmp/prof_d && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -o d -- python $GRAFT_REPO_ROOT/tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 10 --warmup 3 > /dev/null 2>&1 )
f=$(find /tmp/prof_d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/final_ddpm_kernel_stats.csv
rm -rf /tmp/prof_s && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/bench_sd.py --bf16 --steps 4 --warmup 2 > /dev/null 2>&1 )
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/final_sd_kernel_stats.csv
python tools/kbench.py --sizes n18 --extra 2>&1 | grep -v amdgpu > gpurun_out/final_kbench_extra.txt
python - <<'PY'
import json
for f in ("final_bench","final_bench_class","final_ddpm","final_sd"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],3), round(d["ms_per_step"],3), d.get("roofline",{}).get("frac"), (d.get("fwd_bwd") or {}).get("frac", (d.get("fwd_bwd") or {}).get("frac_whole_step")))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/final_convbench.txt
