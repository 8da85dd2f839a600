cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_classification_gpu.py tests/test_fullsize_gpu.py tests/test_classwise_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2 3; do
for v in 0 1; do
SALUN_FWD_SHORTCUT_BESIDE=$v timeout 300 python bench.py --no_cpu_baseline --no_ddpm --no_sd --no_dp --steps 177 --warmup 30 > gpurun_out/fs.json 2> gpurun_out/fs.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/fs.json') if l.startswith('{')][-1]); print('shortcut_beside=$v', round(d['value'],2), round(d['ms_per_step'],4))"
done; done
