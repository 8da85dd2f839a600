# Round 5: the whole -m gpu suite, smoke(), and the default bench line once more (its wall time is part of the record).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r05_gpu_suite.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> gpurun_out/r05_gpu_suite.txt 2>&1
cat gpurun_out/r05_gpu_suite.txt
( time timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_net_of_event_overhead"], d["fwd_bwd"]["frac"])
print("mask_gen", d["mask_gen"])
print("ddpm", {k: v for k, v in d["ddpm"].items() if k in ("value", "ms_per_step", "error")})
PY
