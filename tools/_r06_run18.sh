P=$PWD/build_lab/prev/unlearn_saliency_amd/libsalun.so
for i in 1 2; do
echo "== prev"; SALUN_LIB=$P timeout 300 python tools/convring_bench.py --cfgs 0 2>&1 | grep "ring cfg\|igemm"
echo "== new"; timeout 300 python tools/convring_bench.py --cfgs 0 2>&1 | grep "ring cfg\|igemm"
done
echo "== prev ddpm"; SALUN_LIB=$P timeout 300 python tools/convring_bench.py --ddpm --cfgs 0 2>&1 | grep "ring cfg"
echo "== new ddpm"; timeout 300 python tools/convring_bench.py --ddpm --cfgs 0 2>&1 | grep "ring cfg"
