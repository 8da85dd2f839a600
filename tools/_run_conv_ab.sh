set -x
echo "=== A (baseline)"; python tools/convbench.py --no_lib 2>&1 | grep -v amdgpu
echo "=== B (min 2 waves/SIMD)"; SALUN_LIB=$PWD/unlearn_saliency_amd/libsalun_b.so python tools/convbench.py --no_lib 2>&1 | grep -v amdgpu
echo "=== bench A"; python bench.py --no_cpu_baseline --no_mask_gen --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
echo "=== bench B"; SALUN_LIB=$PWD/unlearn_saliency_amd/libsalun_b.so python bench.py --no_cpu_baseline --no_mask_gen --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
echo "=== proximal diag"; python tools/_diag_prox2.py 2>&1 | grep -v amdgpu
