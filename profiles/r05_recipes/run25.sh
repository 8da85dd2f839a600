# Round 5, GPU call 25: the committed K11 backward-weight (paired 3x3 stride-1 kernel, per-lane direct stores, whole-row
# split reduce): parity, the layer table, where its time goes (lab builds e1 = no stores, e2 = no reduction), and the
# SD step against build_lab/base on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_conv_bf16_gpu.py -x -q 2>&1 | tail -2 )
for v in . build_lab/e1 build_lab/e2 build_lab/pair; do
  n=$(basename $v); [ "$n" = "." ] && n=tree
  ( cd $v && timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_wgp_$n.txt
done
paste -d' ' <(awk '{print substr($0,1,36) substr($0,80,16)}' gpurun_out/r05_wgp_tree.txt) <(awk '{print substr($0,80,16)}' gpurun_out/r05_wgp_e1.txt) <(awk '{print substr($0,80,16)}' gpurun_out/r05_wgp_e2.txt) <(awk '{print substr($0,80,16)}' gpurun_out/r05_wgp_pair.txt) | grep -v "^total"
grep -h "total wgrad" gpurun_out/r05_wgp_*.txt
for rep in 1 2; do
  timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree', round(d['value'],3), round(d['ms_per_step'],2))"
  ( cd build_lab/base && timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base     ', round(d['value'],3), round(d['ms_per_step'],2))" )
done
