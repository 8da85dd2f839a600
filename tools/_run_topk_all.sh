set -x
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_next_gpu.py -x -q -k "topk or smoke or proximal or mask" --timeout 600 2>&1 | tail -8
timeout 600 python tools/topk_scale.py 2>&1 | tail -8
timeout 600 python tools/kbench.py --sizes n18,nd --iters 30 --extra 2>&1 | grep -i 'proximal\|mask_topk'
for cfg in "n18 1" "n18 10" "nd 1" "ns 1"; do
  set -- $cfg
  KEEP_TRACE=0 bash tools/prof.sh topk_$1_$2 python tools/topk_prof.py $1 $2 10 2>&1 | grep -v "^$" | grep "mask_topk n=\|k_\|fill" | cut -c1-160
done
