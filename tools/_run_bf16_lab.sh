# lab: build libsalun variants with one pinned bf16 igemm tile each (build_lab/libsalun_t<N>.so; select with SALUN_LIB)
set -e
cd unlearn_saliency_amd/csrc
make -s
others=$(ls *.o | grep -v salun_conv_bf16.o)
for t in ${TILES:-1 2 3 4}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden -DSALUN_BF16_TILE=$t ${EXTRA:-} -c salun_conv_bf16.hip -o /tmp/cb_$t.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_lab/libsalun_t$t.so $others /tmp/cb_$t.o
done
