# Round 5, after final_d: the default bench line of the final tree once more, on whatever box comes (final_d's box ran
# every kernel 5 - 20 % slower than the round's other boxes: k_masked_adam at N_S 5.02 ms against 4.10 ms).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py > gpurun_out/r05_bench_box2.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_box2.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_net_of_event_overhead'], d['fwd_bwd']['frac'], d['ddpm'].get('value'))"
