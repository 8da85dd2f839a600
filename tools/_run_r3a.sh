set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_classwise_gpu.py tests/test_rccl_ws1_gpu.py tests/test_next_gpu.py tests/test_attn_gpu.py tests/test_kernels_gpu.py -x -q -m gpu --timeout 600 -s 2>&1 | grep -v "^Replacing\|Dataset information\|images for testing\|setup random" | tail -40 > gpurun_out/r3a_tests.txt
tail -15 gpurun_out/r3a_tests.txt
timeout 300 python bench.py --steps 177 --warmup 10 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; tail -c 600 gpurun_out/r3a_bench.err
timeout 300 python bench.py --forget class --no_cpu_baseline > gpurun_out/r3a_bench_class.json 2>> gpurun_out/r3a_bench.err
timeout 400 python bench.py --workload ddpm --no_cpu_baseline > gpurun_out/r3a_ddpm.json 2> gpurun_out/r3a_ddpm.err; tail -c 400 gpurun_out/r3a_ddpm.err
timeout 500 python bench.py --workload sd > gpurun_out/r3a_sd.json 2> gpurun_out/r3a_sd.err; tail -c 400 gpurun_out/r3a_sd.err
python - <<'PY'
import json
for f in ("r3a_bench","r3a_bench_class","r3a_ddpm","r3a_sd"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d.get("ms_per_step"), d.get("roofline",{}).get("frac"), d.get("fwd_bwd",{}).get("frac"), d.get("samples_per_sec"))
    except Exception as e:
        print(f, "ERR", e)
PY
