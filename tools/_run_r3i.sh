cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/wgradbench.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_conv_gpu.py "tests/test_fullsize_gpu.py::test_conv_kernels_at_the_bench_batch_sizes" tests/test_f4_gpu.py -x -q -m gpu -s 2>&1 | grep "of scale\|passed\|failed\|Error" | tail -12
python bench.py --no_cpu_baseline --steps 177 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ddpm', d['value'], d['ms_per_step'], d['fwd_bwd']['frac_whole_step'])"
