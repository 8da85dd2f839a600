cd /tmp && export TMPDIR=/tmp
export MASTER_PORT=29633
R=$GRAFT_REPO_ROOT
rm -rf /tmp/qt; mkdir -p /tmp/qt
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/qt -o dp -- python $R/bench.py --gpus 1 --force_collectives --workload sd --steps 3 --warmup 1 --no_cpu_baseline ) > /tmp/qt/run.log 2>&1
f=$(find /tmp/qt -name "*kernel_trace.csv" | head -1)
head -1 $f | cut -c1-400
python $R/tools/queue_timeline.py $f > $R/gpurun_out/r06_sd_dp_queues.txt 2>&1
cat $R/gpurun_out/r06_sd_dp_queues.txt | head -60
