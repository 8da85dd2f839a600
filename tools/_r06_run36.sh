cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sd_gpu.py tests/test_sd_parity_gpu.py tests/test_norm_bf16_gpu.py tests/test_tok_bf16_gpu.py tests/test_conv_bf16_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15
timeout 600 python bench.py --workload sd --steps 5 --warmup 2 --no_cpu_baseline > gpurun_out/sd_res.json 2> gpurun_out/sd_res.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/sd_res.json') if l.startswith('{')][-1]); print(round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1), round(d['hbm_peak_alloc_GB'],1)); print(d.get('resident_activations'))"
tail -3 gpurun_out/sd_res.err
