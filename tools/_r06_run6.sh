NR=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so
echo "== product (ring wgrad)"; timeout 300 python tools/sustained_bench.py 2>&1 | grep -v amdgpu.ids
echo "== wgrad_v"; SALUN_LIB=$NR timeout 300 python tools/sustained_bench.py 2>&1 | grep -v amdgpu.ids
