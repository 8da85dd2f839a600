# Round 5, GPU call 22: K11 backward-weight at TWO waves per SIMD with one register set (build_lab/wg2,
# -DSALUN_BF16_WGRAD_OCC2=1) against the shipped one-wave three-set ring; the SD layer table, alternated on one box,
# then the backward-weight parity tests on the lab build.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
  timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_wg_ring_$r.txt
  ( cd build_lab/wg2 && timeout 600 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_wg_occ2_$r.txt
done
paste -d'\n' gpurun_out/r05_wg_ring_1.txt gpurun_out/r05_wg_occ2_1.txt | cut -c1-150
grep -h "total wgrad" gpurun_out/r05_wg_ring_2.txt gpurun_out/r05_wg_occ2_2.txt
