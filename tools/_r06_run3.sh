mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_conv_ring_gpu.py -x -q > gpurun_out/r06/t3.txt 2>&1
tail -3 gpurun_out/r06/t3.txt
for i in 1 2; do
SALUN_LIB=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so timeout 200 python tools/convbench.py --no_lib > gpurun_out/r06/cb3_noring_$i.txt 2>&1
timeout 200 python tools/convbench.py --no_lib > gpurun_out/r06/cb3_ring_$i.txt 2>&1
done
SALUN_LIB=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so timeout 200 python tools/convbench.py --no_lib --ddpm > gpurun_out/r06/cb3d_noring.txt 2>&1
timeout 200 python tools/convbench.py --no_lib --ddpm > gpurun_out/r06/cb3d_ring.txt 2>&1
for i in 1 2; do
SALUN_LIB=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen > gpurun_out/r06/b3_noring_$i.json 2>gpurun_out/r06/b3_err.txt
timeout 200 python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen > gpurun_out/r06/b3_ring_$i.json 2>>gpurun_out/r06/b3_err.txt
done
cat gpurun_out/r06/cb3_noring_2.txt gpurun_out/r06/cb3_ring_2.txt
for f in gpurun_out/r06/b3_*.json; do echo $f; python -c "
import json,sys
for ln in open('$f'):
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); print(d.get('value'), d.get('ms_per_step'), d.get('unit'))
"; done
