cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NR=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so
SALUN_LIB=$NR KEEP_TRACE=1 timeout 300 bash tools/prof.sh r06a_wv python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_ddpm --no_mask_gen > /dev/null 2>&1
KEEP_TRACE=1 timeout 300 bash tools/prof.sh r06a_wr python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_ddpm --no_mask_gen > /dev/null 2>&1
head -16 gpurun_out/r06a_wv_kernel_stats.csv | cut -c1-150
head -16 gpurun_out/r06a_wr_kernel_stats.csv | cut -c1-150
ls -la gpurun_out/
