# round 4, GPU call C: kernel-stats profile of the SD bf16 step with the Linear layers on K16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_s && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/bench_sd.py --bf16 --own_linear --steps 4 --warmup 2 > /tmp/sd_prof.json 2>/tmp/sd_prof.err )
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r4c_sd_k16_kernel_stats.csv; tail -1 /tmp/sd_prof.json | cut -c1-300
timeout 900 python -m pytest tests/test_rccl_ws1_gpu.py tests/test_fullsize_diffusion_gpu.py tests/test_gemm_gpu.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -5
