cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k on_vs_off 2>&1 | tail -3
