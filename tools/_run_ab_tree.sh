# Same-box A/B of two TREES (kernel changes of ~1 % drown in the box-to-box spread, and a change can cost the step
# time without showing in any kernel timed alone — DESIGN.md §6).  Before calling:
#   mkdir -p build_lab/old && git archive <older-commit> | tar -x -C build_lab/old && make -C build_lab/old/unlearn_saliency_amd/csrc
#   gpurun -- 'bash tools/_run_ab_tree.sh'        (build_lab/ travels with the snapshot; remove it afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  ( cd build_lab/old && python bench.py --steps 177 --warmup 10 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])" )
  python bench.py --steps 177 --warmup 10 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'], d['fwd_bwd']['frac'])"
done
