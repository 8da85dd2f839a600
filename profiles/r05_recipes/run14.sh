# Round 5, GPU call 14: kernel timeline of the DDPM unlearning step (same reading as profiles/r05_resnet_timeline.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KEEP_TRACE=1 timeout 600 bash tools/prof.sh r05k_ddpm python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 1 --steps 4 --warmup 2 > /dev/null 2>&1
ls -la gpurun_out/r05k_ddpm_trace_slim.csv
