# Round 6, the record of the final tree: default bench line (+ wall time), the -m gpu suite + smoke(), the SD workload line
# and its rocprofv3 kernel table.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ) 2>&1 | tail -3 > gpurun_out/r06_bench_wall.txt
( time timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 ) > gpurun_out/r06_gpu_suite.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> gpurun_out/r06_gpu_suite.txt 2>&1
timeout 900 python bench.py --workload sd --steps 8 --warmup 2 > gpurun_out/r06_sd_bench_bf16.json 2> gpurun_out/r06_sd_bench.err
bash tools/prof.sh r06_sd_bf16 python tools/bench_sd.py --bf16 --steps 8 --warmup 2 --no_cpu_baseline > gpurun_out/r06_sd_prof_head.txt 2>&1
cat gpurun_out/r06_bench_wall.txt gpurun_out/r06_gpu_suite.txt
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench.json") if l.startswith("{")][-1])
print("bench", round(d["value"], 2), round(d["ms_per_step"], 3), d["roofline"]["frac"], d["roofline"]["frac_net_of_event_overhead"], d["fwd_bwd"]["frac"], d["roofline"]["mean_launch_us"])
for k in ("ddpm", "sd"):
    print(" ", k, {a: b for a, b in d[k].items() if a in ("value", "ms_per_step", "error", "host_enqueue_ms_per_step")}, (d[k].get("cpu_baseline") or {}).get("value"), d[k].get("roofline", {}).get("frac"), d[k].get("fwd_bwd"))
r = d["sd"].get("resident_activations") or {}
print("  sd resident", {a: r.get(a) for a in ("value", "ms_per_step", "host_enqueue_ms_per_step", "hbm_peak_alloc_GB")}, (r.get("fwd_bwd") or {}).get("frac"))
print("  dp_ws1", {k: (v.get("ms_per_step"), v.get("dp_over_plain")) for k, v in d["dp_ws1"].items() if isinstance(v, dict)})
print("  cpu", d["cpu_baseline"]["value"], d["mask_gen"]["topk_10_thresholds_device_us"], d["mask_gen"]["total_sec"])
s = json.loads([l for l in open("gpurun_out/r06_sd_bench_bf16.json") if l.startswith("{")][-1])
print("sd line", round(s["value"], 3), round(s["ms_per_step"], 2), s["host_enqueue_ms_per_step"], s["roofline"]["frac"], s["fwd_bwd"]["frac"], (s.get("resident_activations") or {}).get("value"), (s.get("cpu_baseline") or {}).get("value"))
PY
head -32 gpurun_out/r06_sd_prof_head.txt | cut -c1-160
