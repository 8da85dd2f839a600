# same-box A/B: k_finish workgroups per threshold (1 vs automatic), three alternations, wall time of the whole call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for gf in 1 0; do for cfg in "n18 1" "nd 1" "ns 1" "n18 10"; do
  set -- $cfg
  echo -n "gf=$gf "; SALUN_TOPK_GF=$gf timeout 300 python tools/topk_prof.py $1 $2 30 2>&1 | grep "mask_topk n=" | cut -c1-60
done; done; done
