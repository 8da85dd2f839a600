# Round 5, GPU call 24: K11 backward-weight — whole-row output (the one-split kernel's tile and the split reduce leave
# through LDS as the consecutive OIHW floats they are) against build_lab/pair (the paired kernel with the per-lane
# 36-byte-stride stores), one box, alternated; parity suites first.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_conv_bf16_gpu.py tests/test_sd_parity_gpu.py -x -q 2>&1 | tail -3 )
for r in 1 2; do
  timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_wge_rows_$r.txt
  ( cd build_lab/pair && timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_wge_pair_$r.txt
done
paste -d'\n' gpurun_out/r05_wge_rows_1.txt gpurun_out/r05_wge_pair_1.txt | awk '{print substr($0,1,36) substr($0,80,40)}' | grep -v "^total [fd]"
grep -h "total wgrad" gpurun_out/r05_wge_rows_2.txt gpurun_out/r05_wge_pair_2.txt
for rep in 1 2; do
  timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree', round(d['value'],3), round(d['ms_per_step'],2))"
  ( cd build_lab/base && timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base     ', round(d['value'],3), round(d['ms_per_step'],2))" )
done
