set -x
timeout 700 python tools/bench_sd.py --bf16 --steps 5 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r02_sd_bench_bf16.json
timeout 400 python tools/convbench_bf16.py --lib --iters 10 2>&1 | grep -v amdgpu > gpurun_out/r02_convbench_bf16.txt
timeout 300 python tools/attnbench.py 2>&1 | grep -v amdgpu > gpurun_out/r02_attnbench.txt
KEEP_TRACE=0 bash tools/prof.sh r02_sd_bf16 python tools/bench_sd.py --bf16 --steps 2 --warmup 1 --mask_batches 1 > /dev/null 2>&1
