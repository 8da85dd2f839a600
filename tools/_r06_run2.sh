mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_conv_ring_gpu.py tests/test_conv_gpu.py -x -q > gpurun_out/r06/t2.txt 2>&1
for i in 1 2; do
SALUN_RING=0 timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen > gpurun_out/r06/b2_ring0_$i.json 2>gpurun_out/r06/b2_err.txt
SALUN_RING=1 timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen > gpurun_out/r06/b2_ring1_$i.json 2>>gpurun_out/r06/b2_err.txt
done
SALUN_RING=0 timeout 300 python tools/bench_ddpm.py --steps 6 --warmup 2 > gpurun_out/r06/d2_ring0.json 2>>gpurun_out/r06/b2_err.txt
SALUN_RING=1 timeout 300 python tools/bench_ddpm.py --steps 6 --warmup 2 > gpurun_out/r06/d2_ring1.json 2>>gpurun_out/r06/b2_err.txt
tail -5 gpurun_out/r06/t2.txt
for f in gpurun_out/r06/b2_ring*.json gpurun_out/r06/d2_ring*.json; do echo $f; python -c "
import json,sys
for ln in open('$f'):
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); print(d.get('value'), d.get('ms_per_step'), d.get('unit'))
"; done
