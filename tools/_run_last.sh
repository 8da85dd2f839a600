cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/last_bench.json 2> gpurun_out/last_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/last_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['fwd_bwd']['frac'], d['step_ms_trend'], d['cpu_baseline']['value'])"
