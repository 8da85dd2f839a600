set -x
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -k "topk or smoke" --timeout 600 2>&1 | tail -5
timeout 600 python tools/topk_scale.py 2>&1 | tail -8
for cfg in "n18 1" "n18 10" "nd 1" "ns 1"; do
  set -- $cfg
  KEEP_TRACE=0 bash tools/prof.sh topk_$1_$2 python tools/topk_prof.py $1 $2 10 > /dev/null 2>&1
done
timeout 2400 python -m pytest tests/test_fullsize_gpu.py -q --timeout 900 -s -k "ft_l1" 2>&1 | grep -v "^$" | grep "passed\|failed\|Error\|l1 comp\|FT_l1 grad" | cut -c1-400
timeout 900 python -m pytest tests/test_sd_parity_gpu.py -q --timeout 900 -k "proximal" 2>&1 | tail -5
