cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_sd.py --bf16 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-300
timeout 600 python tools/bench_sd.py --bf16 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-300
