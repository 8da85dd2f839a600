cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
B="python bench.py --no_cpu_baseline --no_mask_gen --steps 177"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B 2>/dev/null | python -c "$P" w8
$B --main_priority -1 2>/dev/null | python -c "$P" w8_mainhigh
SALUN_LIB=$PWD/build_lab/libsalun_w0.so $B 2>/dev/null | python -c "$P" w4
SALUN_LIB=$PWD/build_lab/libsalun_w0.so $B --main_priority -1 2>/dev/null | python -c "$P" w4_mainhigh
python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | python -c "$P" ddpm_w8
SALUN_LIB=$PWD/build_lab/libsalun_w0.so python bench.py --workload ddpm --no_cpu_baseline 2>/dev/null | python -c "$P" ddpm_w4
