# Round 6: with backward-weight on the side stream, are the reduction-split targets of rounds 3 - 5 (tuned with everything
# on one stream) still right?  Lab builds (tools/lab_build.sh flag <name> <file> -D...) selected through SALUN_LIB, one box.
#   tn128 ... tn1024: -DSALUN_TN_TARGET=...  (salun_gemm.hip: the Linear layers' dY^T.X; 512 until this experiment, now 256)
#   wg192 ... wg768:  -DSALUN_BF16_WGRAD_TARGET=... (salun_conv_bf16.hip, default 384: 3x3 backward-weight)
#   sp384 ... sp1024: -DSALUN_BF16_SPLIT_TARGET=... (salun_conv_bf16.hip, default 768: forward / backward-data reduction split)
# First pass (tree = tn512): tn256 -0.8 % / -1.9 % (resident), tn128 mixed, tn1024 +1.2 %, wg192 0 / -1 %, wg768 +2 %:
# profiles/r06_sd_split_targets.txt holds both passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() {
  label=$1; lib=$2
  if [ -n "$lib" ]; then export SALUN_LIB=$GRAFT_REPO_ROOT/build_lab/$lib/unlearn_saliency_amd/libsalun.so; else unset SALUN_LIB; fi
  timeout 600 python bench.py --workload sd --steps 5 --warmup 2 --no_cpu_baseline > gpurun_out/sd_ab.json 2> gpurun_out/sd_ab.err
  python - "$label" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/sd_ab.json") if l.startswith("{")][-1]); r = d.get("resident_activations") or {}
print(sys.argv[1], round(d["value"], 3), round(d["ms_per_step"], 2), "| resident", round(r.get("value"), 3), round(r.get("ms_per_step"), 2))
PY
}
for i in 1 2; do
  one "tree (tn256 wg384 sp768)" ""
  one "tn192                   " tn192
  one "tn384                   " tn384
  one "wg192                   " wg192
  one "wg256                   " wg256
  one "sp512                   " sp512
  one "sp384                   " sp384
  one "sp1024                  " sp1024
done
