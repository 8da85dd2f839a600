# Round 5, the measurements DESIGN.md / profiles/README.md quote: bench lines, rocprofv3 kernel tables, kernel
# micro-benchmarks, top-k per-kernel tables (+ the round-3/4 top-k on the SAME box: build_lab/r03 = `git archive bff0ec6`
# + make), PMC traffic passes (counters in their own runs).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
timeout 300 python bench.py --forget class --no_cpu_baseline --no_ddpm > gpurun_out/r05_bench_class.json 2>/dev/null
timeout 600 python bench.py --workload ddpm --no_cpu_baseline > gpurun_out/r05_ddpm_bench.json 2>/dev/null
timeout 900 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline > gpurun_out/r05_sd_bench_bf16.json 2>/dev/null
KEEP_TRACE=0 timeout 600 bash tools/prof.sh r05_bench python bench.py --no_cpu_baseline --no_ddpm --steps 177 > /dev/null 2>&1
KEEP_TRACE=0 timeout 600 bash tools/prof.sh r05_ddpm python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 10 --warmup 3 > /dev/null 2>&1
timeout 900 python tools/kbench.py --sizes n18,nd,ns --iters 30 --extra --json gpurun_out/r05_kbench.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_kbench.txt
for cfg in "n18 1" "n18 10" "nd 1" "nd 10" "ns 1"; do
  set -- $cfg
  KEEP_TRACE=0 timeout 300 bash tools/prof.sh r05_topk_$1_$2 python tools/topk_prof.py $1 $2 10 > /dev/null 2>&1
done
( for rep in 1 2; do for cfg in "n18 1" "n18 10" "nd 1" "nd 10" "ns 1"; do
    set -- $cfg
    echo -n "r05 tree  "; timeout 300 python tools/topk_prof.py $1 $2 30 2>&1 | grep "mask_topk n=" | cut -c1-100
    echo -n "r03 tree  "; ( cd build_lab/r03 && timeout 300 python tools/topk_prof.py $1 $2 30 2>&1 | grep "mask_topk n=" | cut -c1-100 )
  done; done ) > gpurun_out/r05_topk_same_box.txt 2>&1
for sz in "n18 11173962" "nd 38632323" "ns 859520964"; do
  set -- $sz
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 400 bash tools/pmc.sh r05_$1 $ctr python tools/kbench_update.py $2 > /dev/null 2>&1
  done
done
cat gpurun_out/r05_topk_same_box.txt
python - <<'PY'
import json
for f in ("r05_bench", "r05_bench_class", "r05_ddpm_bench", "r05_sd_bench_bf16"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 3), round(d["ms_per_step"], 3), (d.get("roofline") or {}).get("frac"),
              (d.get("fwd_bwd") or {}).get("frac", (d.get("fwd_bwd") or {}).get("frac_whole_step")))
        if f == "r05_bench":
            print("  mask_gen", d["mask_gen"]); print("  roofline", d["roofline"]); print("  ddpm", {k: v for k, v in d["ddpm"].items() if k in ("value", "ms_per_step", "error")})
            print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cpu_model"], d["cpu_baseline"]["sample"][:60])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -c 600 gpurun_out/r05_bench.err
