cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/igemmbench.py 2>&1 | grep -v amdgpu
SALUN_LIB=$PWD/build_lab/libsalun_il0.so python tools/igemmbench.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q -m gpu 2>&1 | tail -2
