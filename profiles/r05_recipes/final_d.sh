# Round 5, last GPU call (after the paired backward-weight kernel and the epilogue work): the whole -m gpu suite and
# smoke() on the final tree, the default bench line, the DDPM / SD / class-wise lines, the bf16 layer table, and the
# rocprofv3 kernel summaries of the ResNet-18, DDPM and SD commands.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 ) > gpurun_out/r05_gpu_suite.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> gpurun_out/r05_gpu_suite.txt 2>&1
cat gpurun_out/r05_gpu_suite.txt
( time timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err ) 2>&1 | tail -3
timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline > gpurun_out/r05_sd_bench_bf16.json 2>/dev/null
timeout 600 python bench.py --workload ddpm --no_cpu_baseline > gpurun_out/r05_ddpm_bench.json 2>/dev/null
timeout 600 python bench.py --forget class --no_cpu_baseline --no_ddpm > gpurun_out/r05_bench_class.json 2>/dev/null
timeout 300 python tools/convbench_bf16.py --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_convbench_bf16_final.txt
timeout 600 bash tools/prof.sh r05_bench python bench.py --no_cpu_baseline --no_ddpm > /dev/null 2>&1
timeout 600 bash tools/prof.sh r05_ddpm python bench.py --workload ddpm --no_cpu_baseline > /dev/null 2>&1
timeout 600 bash tools/prof.sh r05_sd_bf16 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_net_of_event_overhead"], d["fwd_bwd"]["frac"])
print("mask_gen", {k: v for k, v in d["mask_gen"].items() if k != "topk_note"})
print("ddpm-in-line", {k: v for k, v in d["ddpm"].items() if k in ("value", "ms_per_step", "error")})
for f in ("r05_sd_bench_bf16", "r05_ddpm_bench", "r05_bench_class"):
    s = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, s["value"], s["ms_per_step"])
PY
tail -3 gpurun_out/r05_convbench_bf16_final.txt
head -4 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-160
