#!/bin/bash
# Second builds of the library for same-box A/B measurements (kernel times differ by 5 - 10 % between GPU boxes, so a
# variant is only ever compared with the tree on ONE box: the recipe runs `python tools/...` here and in build_lab/<name>).
# build_lab/ is git-ignored and travels to the GPU box with the snapshot.
#   tools/lab_build.sh rev  <name> <git-rev>                 a full build of a revision (e.g. `base` = the tree before a change)
#   tools/lab_build.sh flag <name> <file-stem> -DMACRO=1 …   this working tree with one source recompiled with extra flags
#                                                            (e.g. flag wg2 salun_conv_bf16 -DSALUN_BF16_WGRAD_TARGET=512)
set -e
mode=$1; name=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root"
rm -rf build_lab/$name; mkdir -p build_lab/$name
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden"
if [ "$mode" = rev ]; then
  git archive "$1" | tar -x -C build_lab/$name
  make -C build_lab/$name/unlearn_saliency_amd/csrc -j8 > /dev/null
  rm -rf build_lab/$name/tests/golden build_lab/$name/oracle
else
  stem=$1; shift
  git archive HEAD unlearn_saliency_amd tools include bench.py BASELINE.json | tar -x -C build_lab/$name
  mkdir -p build_lab/$name/profiles; cp profiles/r05_pmc_traffic.json build_lab/$name/profiles/ 2>/dev/null || true
  cp unlearn_saliency_amd/csrc/*.hip unlearn_saliency_amd/csrc/*.h unlearn_saliency_amd/csrc/*.o build_lab/$name/unlearn_saliency_amd/csrc/
  ( cd build_lab/$name/unlearn_saliency_amd/csrc && /opt/rocm/bin/hipcc $FLAGS "$@" -c $stem.hip -o $stem.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsalun.so *.o )
fi
rm -f build_lab/$name/unlearn_saliency_amd/csrc/*.o
