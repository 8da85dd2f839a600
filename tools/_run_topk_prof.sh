set -x
for cfg in "n18 1" "n18 10" "nd 1" "ns 1"; do
  set -- $cfg
  KEEP_TRACE=0 bash tools/prof.sh topk_$1_$2 python tools/topk_prof.py $1 $2 10 2>&1 | grep -v "^$" | grep "mask_topk n=\|k_\|fill" | cut -c1-160
done
