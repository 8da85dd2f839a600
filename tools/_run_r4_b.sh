# round 4, GPU call B: re-run what failed in A, HIP-graph capture of the SD step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
( timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py -k "grouped or attention or ddpm_unet or device_resident" -q --timeout 600 -p no:cacheprovider 2>&1 | tail -80 ) > gpurun_out/r4b_tests_kernels.txt
( timeout 1500 python -m pytest tests/test_ddpm_gpu.py tests/test_dist_diffusion_gpu.py tests/test_rccl_ws1_gpu.py tests/test_sd_parity_gpu.py tests/test_f4_gpu.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -120 ) > gpurun_out/r4b_tests_ddpm.txt
SALUN_OWN_GEMM=1 timeout 600 python tools/bench_ddpm.py --steps 10 --warmup 3 --mask_batches 4 --no_cpu_baseline > gpurun_out/r4b_ddpm_owngemm1.json 2> gpurun_out/r4b_ddpm_owngemm1.err
timeout 900 python tools/bench_sd.py --bf16 --steps 4 --warmup 1 --graph > gpurun_out/r4b_sd_lib_graph.json 2> gpurun_out/r4b_sd_lib_graph.err
timeout 900 python tools/bench_sd.py --bf16 --steps 4 --warmup 1 --graph --own_linear > gpurun_out/r4b_sd_k16_graph.json 2> gpurun_out/r4b_sd_k16_graph.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4b_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", round(d["value"], 3), "ms", round(d["ms_per_step"], 2), "host", round(d.get("host_enqueue_ms_per_step", 0), 1), d.get("hip_graph"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-1500:])
PY
tail -15 gpurun_out/r4b_tests_kernels.txt; tail -30 gpurun_out/r4b_tests_ddpm.txt
