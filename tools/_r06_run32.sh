cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KEEP_TRACE=1 timeout 300 bash tools/prof.sh r06d_dp python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_ddpm --no_mask_gen --no_sd --no_dp --force_collectives > /dev/null 2>&1
python tools/step_timeline.py gpurun_out/r06d_dp_trace_slim.csv 3000 | head -70
