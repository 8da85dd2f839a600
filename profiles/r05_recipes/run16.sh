# Round 5, GPU call 16: K11 forward / backward-data with descriptor-based staging (no per-stage vector-ALU address
# work, conflict-free LDS stores) — parity tests, the SD layer table for this tree and for build_lab/base (= HEAD before
# the change) on the same box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_conv_bf16_gpu.py -x -q 2>&1 | tail -4 )
timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_convbench_bf16.txt
( cd build_lab/base && timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_convbench_bf16_base.txt
paste -d'\n' gpurun_out/r05_convbench_bf16.txt gpurun_out/r05_convbench_bf16_base.txt | cut -c1-150
