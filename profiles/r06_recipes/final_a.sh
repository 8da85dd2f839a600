# Round 6, the measurements DESIGN.md / profiles/README.md quote.  Part A: the default bench line (wall time recorded),
# per-workload lines, rocprofv3 kernel tables, layer tables at the sustained clock.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ) 2>&1 | tail -3 > gpurun_out/r06_bench_wall.txt
timeout 300 python bench.py --forget class --no_cpu_baseline --no_ddpm --no_sd --no_dp > gpurun_out/r06_bench_class.json 2>/dev/null
timeout 600 python bench.py --workload ddpm --no_cpu_baseline > gpurun_out/r06_ddpm_bench.json 2>/dev/null
timeout 900 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline > gpurun_out/r06_sd_bench_bf16.json 2>/dev/null
KEEP_TRACE=1 timeout 600 bash tools/prof.sh r06_bench python bench.py --no_cpu_baseline --no_ddpm --no_sd --no_dp --steps 177 > /dev/null 2>&1
python tools/step_timeline.py gpurun_out/r06_bench_trace_slim.csv 1e18 > gpurun_out/r06_resnet_timeline.txt 2>&1
KEEP_TRACE=0 timeout 600 bash tools/prof.sh r06_ddpm python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 10 --warmup 3 > /dev/null 2>&1
KEEP_TRACE=0 timeout 900 bash tools/prof.sh r06_sd_bf16 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
( timeout 400 python tools/convbench.py --no_lib 2>&1 | grep -v amdgpu.ids; timeout 400 python tools/convbench.py --no_lib --ddpm 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_convbench.txt
( timeout 400 python tools/convring_bench.py --cfgs 0,1,2,3,5 2>&1 | grep -v amdgpu.ids; timeout 400 python tools/convring_bench.py --ddpm --cfgs 0,1,2,3 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_convring_ab.txt
cat gpurun_out/r06_bench_wall.txt gpurun_out/r06_convbench.txt
python - <<'PY'
import json
for f in ("r06_bench", "r06_bench_class", "r06_ddpm_bench", "r06_sd_bench_bf16"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["value"], 3), round(d["ms_per_step"], 3), (d.get("roofline") or {}).get("frac"),
              (d.get("fwd_bwd") or {}).get("frac"))
        if f == "r06_bench":
            print("  mask_gen", d["mask_gen"]); print("  roofline", d["roofline"])
            for k in ("ddpm", "sd"):
                print(" ", k, {a: b for a, b in d[k].items() if a in ("value", "ms_per_step", "error")}, (d[k].get("cpu_baseline") or {}).get("value"))
            print("  dp_ws1", {k: (v.get("ms_per_step"), v.get("dp_over_plain")) for k, v in d["dp_ws1"].items() if isinstance(v, dict)})
            print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cpu_model"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -c 400 gpurun_out/r06_bench.err
