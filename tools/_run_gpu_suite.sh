cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -15 > gpurun_out/suite.txt; cat gpurun_out/suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 177 --warmup 10 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/final_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['fwd_bwd']['frac'], d['cpu_baseline']['value'], d['mask_gen_sec'])"
