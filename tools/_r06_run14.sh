cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KEEP_TRACE=1 timeout 300 bash tools/prof.sh r06b_dp python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_ddpm --no_mask_gen --force_collectives > /dev/null 2>&1
python tools/step_timeline.py gpurun_out/r06b_dp_trace_slim.csv 7000
