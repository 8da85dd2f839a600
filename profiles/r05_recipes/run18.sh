# Round 5, GPU call 18: K11 tile shapes after the staging change (SALUN_BF16_TILE lab builds): default routing vs
# 256 x 128 (eight waves) vs 128 x 256 everywhere, same box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in . build_lab/tile5 build_lab/tile6; do
  echo "== $v"; ( cd $v && timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-100 )
done
