# Round-4 measurements, second pass (after the 1x1 ring convolution and the host-side changes): the bench lines and kernel
# stats again, the K11 layer table, and a traffic pass over the whole DDPM step (every kernel of rounds 3 and 4 it runs).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python bench.py --forget class --no_cpu_baseline > gpurun_out/r04_bench_class.json 2>/dev/null
python bench.py --workload ddpm > gpurun_out/r04_ddpm_bench.json 2>/dev/null
python bench.py --workload sd --steps 6 --warmup 2 > gpurun_out/r04_sd_bench_bf16.json 2>/dev/null
python tools/bench_sd.py --steps 3 --warmup 1 > gpurun_out/r04_sd_bench_f32.json 2>/dev/null
prof() {
  tag=$1; shift
  rm -rf /tmp/prof_$tag && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" > /dev/null 2>&1 )
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04_${tag}_kernel_stats.csv
}
prof bench python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --steps 177
prof ddpm python $GRAFT_REPO_ROOT/tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 10 --warmup 3
prof sd_bf16 python $GRAFT_REPO_ROOT/tools/bench_sd.py --bf16 --steps 5 --warmup 2
python tools/convbench_bf16.py 2>&1 | grep -v amdgpu > gpurun_out/r04_convbench_bf16.txt
bash tools/pmc.sh r04_ddpmstep FETCH_SIZE python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/pmc.sh r04_ddpmstep WRITE_SIZE python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 1 --steps 3 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import json
for f in ("r04_bench","r04_bench_class","r04_ddpm_bench","r04_sd_bench_bf16","r04_sd_bench_f32"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],3), round(d["ms_per_step"],3), d.get("roofline",{}).get("frac"), (d.get("fwd_bwd") or {}).get("frac", (d.get("fwd_bwd") or {}).get("frac_whole_step")), (d.get("roofline") or {}).get("traffic_source","")[:40])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r04_convbench_bf16.txt; head -12 gpurun_out/r04_ddpmstep_FETCH_SIZE.csv | cut -c1-120
