cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_sd_parity_gpu.py::test_proximal_gradient_on_device_vs_the_reference_run tests/test_f4_gpu.py tests/test_fullsize_diffusion_gpu.py tests/test_ddpm_gpu.py -q -m gpu --timeout 900 2>&1 | tail -8
rm -rf /tmp/prof_d && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -o d -- python $GRAFT_REPO_ROOT/bench.py --workload ddpm --no_cpu_baseline --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r3h_ddpm_prof.json 2>/dev/null )
f=$(find /tmp/prof_d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3h_ddpm_kernel_stats.csv
rm -rf /tmp/prof_s && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/bench.py --workload sd --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3h_sd_prof.json 2>/dev/null )
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3h_sd_kernel_stats.csv
python bench.py --workload ddpm --steps 20 --warmup 3 > gpurun_out/r3h_ddpm.json 2>/dev/null
python bench.py --workload sd --steps 5 --warmup 2 > gpurun_out/r3h_sd.json 2>/dev/null
python - <<'PY'
import json
for f in ("r3h_ddpm","r3h_sd","r3h_ddpm_prof","r3h_sd_prof"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d.get("ms_per_step"), d.get("roofline",{}).get("frac"), d.get("fwd_bwd",{}), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
