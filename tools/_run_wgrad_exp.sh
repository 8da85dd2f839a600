cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/wgradbench.py 2>&1 | grep -v amdgpu
for e in "$@"; do SALUN_LIB=$PWD/build_lab/libsalun_exp$e.so python tools/wgradbench.py 2>&1 | grep -v amdgpu; done
