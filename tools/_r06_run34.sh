cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_sd_gpu.py tests/test_sd_parity_gpu.py tests/test_fullsize_arith_gpu.py tests/test_fullsize_diffusion_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5
for i in 1 2; do
for bp in 0 1; do
SALUN_SD_EMB_FP32=$bp timeout 500 python tools/bench_sd.py --bf16 --steps 5 --warmup 2 --no_cpu_baseline > gpurun_out/sd_bp$bp.json 2> gpurun_out/sd_bp$bp.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/sd_bp$bp.json') if l.startswith('{')][-1]); print('emb_fp32=$bp', round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1))"
done; done
