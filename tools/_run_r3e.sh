cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_b && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --steps 177 > $GRAFT_REPO_ROOT/gpurun_out/r3e_bench_prof.json 2>/dev/null )
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3e_bench_kernel_stats.csv; head -14 gpurun_out/r3e_bench_kernel_stats.csv | cut -c1-200
rm -rf /tmp/prof_w && ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o w -- python $GRAFT_REPO_ROOT/tools/wgradbench.py > /dev/null 2>&1 )
f=$(find /tmp/prof_w -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3e_wgrad_kernel_stats.csv; head -8 gpurun_out/r3e_wgrad_kernel_stats.csv | cut -c1-220
