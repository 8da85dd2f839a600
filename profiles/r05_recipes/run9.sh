# Round 5, GPU call 9: kernel timeline of the ResNet-18 step (is the side stream's backward-weight chain the critical
# path of backward?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KEEP_TRACE=1 timeout 300 bash tools/prof.sh r05h_step python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_ddpm --no_mask_gen > /dev/null 2>&1
ls -la gpurun_out/r05h_step_trace_slim.csv
