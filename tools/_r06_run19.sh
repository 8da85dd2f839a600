timeout 600 python tools/convring_bench.py --cfgs 0,1,2,3,5 2>&1 | grep "ring cfg\|igemm"
timeout 600 python tools/convring_bench.py --ddpm --cfgs 0,1,2,3 2>&1 | grep "ring cfg\|igemm"
