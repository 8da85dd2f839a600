cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 768 512 384 1024; do
  SALUN_BF16_SPLIT_WGS=$w timeout 900 python tools/bench_sd.py --bf16 --steps 4 --warmup 2 > gpurun_out/blk5_sd_$w.json 2>gpurun_out/blk5_sd_$w.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/blk5_sd_$w.json').read().strip().splitlines()[-1]); print('sd split wgs=$w', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/blk5_sd_$w.err
done
