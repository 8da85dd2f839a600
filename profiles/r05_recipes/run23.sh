# Round 5, GPU call 23: K11 backward-weight 3x3 stride 1 as the paired variant (two workgroups per CU, one register
# set) in the product tree: parity suites, the split-target sweep (lab builds t512f / t512c / t256 against 384), and
# the SD step against build_lab/base (the tree before this round's K11 changes) on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_conv_bf16_gpu.py tests/test_sd_parity_gpu.py tests/test_sd_gpu.py -x -q 2>&1 | tail -3 )
for r in 1 2; do
  timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_wgt_384_$r.txt
  for v in t512f t512c t256; do
    ( cd build_lab/$v && timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_wgt_${v}_$r.txt
  done
done
for v in 384 t512f t512c t256; do echo "== $v"; awk '{print substr($0,1,36) substr($0,80,40)}' gpurun_out/r05_wgt_${v}_1.txt | grep -v "^total [fd]"; grep -h "total wgrad" gpurun_out/r05_wgt_${v}_2.txt; done
for rep in 1 2; do
  timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree', round(d['value'],3), round(d['ms_per_step'],2))"
  ( cd build_lab/base && timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base     ', round(d['value'],3), round(d['ms_per_step'],2))" )
done
