cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/igemmbench.py 2>&1 | grep -v amdgpu
for e in "$@"; do SALUN_LIB=$PWD/build_lab/libsalun_iexp$e.so python tools/igemmbench.py 2>&1 | grep -v amdgpu; done
