cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_conv_bf16_gpu.py tests/test_sd_gpu.py tests/test_sd_parity_gpu.py tests/test_fullsize_diffusion_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
for i in 1 2; do
timeout 600 python bench.py --workload sd --steps 5 --warmup 2 --no_cpu_baseline > gpurun_out/sd_res.json 2> gpurun_out/sd_res.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/sd_res.json') if l.startswith('{')][-1]); r=d['resident_activations']; print(round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1), '| resident', round(r['value'],3), round(r['ms_per_step'],2), round(r['host_enqueue_ms_per_step'],1), round(r['hbm_peak_alloc_GB'],1))"
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sd -o sd -- python $GRAFT_REPO_ROOT/tools/bench_sd.py --bf16 --steps 8 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_sd -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06_sd_bf16_kernel_stats.csv; head -30 gpurun_out/r06_sd_bf16_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof_sd
