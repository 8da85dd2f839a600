for v in 1 2 4; do
echo "== ring wgrad EXP=$v"; SALUN_LIB=$PWD/build_lab/wgrx$v/unlearn_saliency_amd/libsalun.so timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids
done
echo "== ring wgrad"; timeout 300 python tools/corun_bench.py 2>&1 | grep -v amdgpu.ids
