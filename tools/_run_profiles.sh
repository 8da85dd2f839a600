set -x
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -c 600 gpurun_out/r02_bench.json
bash tools/prof.sh r02_bench python bench.py --no_cpu_baseline --steps 60 | head -30 | cut -c1-160
timeout 900 python tools/kbench.py --sizes n18,nd,ns --iters 30 --extra --json gpurun_out/r02_kbench.json 2>&1 | tail -70
for sz in "n18 11173962" "nd 38632323"; do set -- $sz
  bash tools/pmc.sh r02_$1 FETCH_SIZE python tools/kbench_update.py $2 > /dev/null 2>&1
  bash tools/pmc.sh r02_$1 WRITE_SIZE python tools/kbench_update.py $2 > /dev/null 2>&1
done
bash tools/pmc.sh r02_ns FETCH_SIZE python tools/topk_prof.py ns 1 3 > /dev/null 2>&1
bash tools/pmc.sh r02_ns WRITE_SIZE python tools/topk_prof.py ns 1 3 > /dev/null 2>&1
python tools/pmc_traffic.py r02_n18:11173962 r02_nd:38632323 r02_ns:859520964 > gpurun_out/r02_pmc_traffic.json; head -c 1500 gpurun_out/r02_pmc_traffic.json
timeout 600 python tools/topk_scale.py > gpurun_out/r02_topk_scale.txt 2>&1; cat gpurun_out/r02_topk_scale.txt
for cfg in "n18 1" "n18 10" "nd 1" "ns 1"; do set -- $cfg
  bash tools/prof.sh r02_topk_$1_$2 python tools/topk_prof.py $1 $2 10 > /dev/null 2>&1
done
timeout 900 python bench.py --workload ddpm --steps 20 --warmup 3 > gpurun_out/r02_ddpm_bench.json 2> gpurun_out/r02_ddpm_bench.err; tail -c 1200 gpurun_out/r02_ddpm_bench.json
