# Round 5, GPU call 28: SD step, this tree against build_lab/pre (the revision before the ring epilogues were changed
# — SD does not use the fp32 GEMM, so the ring / K16 epilogue is the only difference), alternated on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],3), round(d['ms_per_step'],3))"; }
for rep in 1 2 3; do
  timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | one "sd this"
  ( cd build_lab/pre && timeout 600 python bench.py --workload sd --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | one "sd pre " )
done
