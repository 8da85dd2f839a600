# lab: build libsalun variants with one pinned bf16 igemm tile each and time the SD layer table with them
set -e
cd unlearn_saliency_amd/csrc
for t in 1 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden -DSALUN_BF16_TILE=$t -c salun_conv_bf16.hip -o /tmp/cb_$t.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_lab/libsalun_t$t.so salun_update.o salun_topk.o salun_loss.o salun_conv.o /tmp/cb_$t.o salun_norm.o salun_prox.o
done
