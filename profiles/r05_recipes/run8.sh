# Round 5, GPU call 8: backward-weight workgroup target sweep (partial-sum traffic vs chip fill) on the ResNet-18 step.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 256 128 192 384 512; do
  SALUN_WGRAD_WGS=$t timeout 300 python bench.py --steps 177 --warmup 10 --no_cpu_baseline --no_ddpm --no_mask_gen 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgs $t', round(d['value'],2), round(d['ms_per_step'],3), round(d['fwd_bwd']['frac'],4), round(d['roofline']['mean_launch_us'],1))"
done
