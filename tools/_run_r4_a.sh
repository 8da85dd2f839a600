# round 4, GPU call A: validate the new kernels / data-parallel paths, first A/B numbers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r04_measured_errors.txt
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
( timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_bf16_gpu.py tests/test_kernels_gpu.py -k "gemm or dropout or softmax or linear or attention or pack or variants or transformer" -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r4a_tests_kernels.txt
( timeout 1500 python -m pytest tests/test_ddpm_block_gpu.py tests/test_target_overlap_gpu.py tests/test_ddpm_gpu.py tests/test_dist_diffusion_gpu.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r4a_tests_ddpm.txt
( timeout 900 python tools/gemmbench_bf16.py --reps 20 2>&1 | grep -v amdgpu ) > gpurun_out/r4a_gemmbench_bf16.txt
for og in 1 0; do
  SALUN_OWN_GEMM=$og timeout 600 python tools/bench_ddpm.py --steps 10 --warmup 3 --mask_batches 4 --no_cpu_baseline > gpurun_out/r4a_ddpm_owngemm$og.json 2> gpurun_out/r4a_ddpm_owngemm$og.err
done
timeout 600 python tools/bench_sd.py --bf16 --steps 3 --warmup 1 > gpurun_out/r4a_sd_lib.json 2> gpurun_out/r4a_sd_lib.err
timeout 600 python tools/bench_sd.py --bf16 --steps 3 --warmup 1 --own_linear > gpurun_out/r4a_sd_k16.json 2> gpurun_out/r4a_sd_k16.err
SALUN_LINEAR_GEMM=0 timeout 600 python tools/bench_sd.py --bf16 --steps 3 --warmup 1 --own_linear > gpurun_out/r4a_sd_k11.json 2> gpurun_out/r4a_sd_k11.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4a_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", round(d["value"], 3), "ms", round(d["ms_per_step"], 2), "host", round(d.get("host_enqueue_ms_per_step", 0), 1))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
tail -5 gpurun_out/r4a_tests_kernels.txt; tail -5 gpurun_out/r4a_tests_ddpm.txt; tail -4 gpurun_out/r4a_gemmbench_bf16.txt
