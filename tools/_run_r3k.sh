cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_gradsink.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_ddpm_gpu.py tests/test_sd_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ddpm', d['value'], d['ms_per_step'], d['fwd_bwd']['frac_whole_step'])"
