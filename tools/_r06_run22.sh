timeout 600 python bench.py --workload sd --steps 4 --warmup 1 --no_cpu_baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sd', d['value'], d['ms_per_step'], 'host enqueue', d['host_enqueue_ms_per_step'], 'adam ms', d['roofline']['mean_launch_ms'])"
timeout 600 python tools/hostprof_diffusion.py sd 2>&1 | grep -v amdgpu | head -60
