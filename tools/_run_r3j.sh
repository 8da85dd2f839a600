cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/igemmbench.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_conv_gpu.py "tests/test_fullsize_gpu.py::test_conv_kernels_at_the_bench_batch_sizes" -x -q -m gpu 2>&1 | tail -2
python bench.py --no_cpu_baseline --steps 177 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
