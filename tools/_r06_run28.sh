for i in 1 2; do
( cd build_lab/r05 && timeout 300 python bench.py --no_cpu_baseline --no_ddpm 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05 tree', round(d['value'],2), round(d['ms_per_step'],3), d['fwd_bwd']['frac'])" )
timeout 300 python bench.py --no_cpu_baseline --no_ddpm --no_sd --no_dp 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r06 tree', round(d['value'],2), round(d['ms_per_step'],3), d['fwd_bwd']['frac'])"
done
( cd build_lab/r05 && timeout 300 python tools/bench_ddpm.py --steps 10 --warmup 3 --mask_batches 2 --no_cpu_baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05 ddpm', round(d['value'],3), round(d['ms_per_step'],2))" )
timeout 300 python tools/bench_ddpm.py --steps 10 --warmup 3 --mask_batches 2 --no_cpu_baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r06 ddpm', round(d['value'],3), round(d['ms_per_step'],2))"
