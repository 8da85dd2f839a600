cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/convbench.py --no_lib > gpurun_out/blk9_convbench.txt 2>&1; tail -13 gpurun_out/blk9_convbench.txt
python bench.py --steps 177 --warmup 10 --no_cpu_baseline > gpurun_out/blk9_bench.json 2> gpurun_out/blk9_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/blk9_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
