cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in "" "--own_linear" "--own_linear --graph"; do
timeout 600 python tools/bench_sd.py --bf16 --steps 4 --warmup 2 $f 2>/tmp/e.txt | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('sd $f', round(d['value'],3), round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],1))"
done
