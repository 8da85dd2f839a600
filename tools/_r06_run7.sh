NR=$PWD/build_lab/noring/unlearn_saliency_amd/libsalun.so
echo "== ring wgrad"; timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids
echo "== wgrad_v"; SALUN_LIB=$NR timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids
