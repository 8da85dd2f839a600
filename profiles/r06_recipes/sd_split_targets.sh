# Round 6: with backward-weight on the side stream, are the reduction-split targets of rounds 3 - 5 (tuned with everything
# on one stream) still right?  Lab builds (tools/lab_build.sh flag <name> <file> -D...) selected through SALUN_LIB, one box.
#   tn128 / tn256 / tn1024: -DSALUN_TN_TARGET=...  (salun_gemm.hip, default 512: the Linear layers' dY^T.X)
#   wg192 / wg768:          -DSALUN_BF16_WGRAD_TARGET=... (salun_conv_bf16.hip, default 384: 3x3 backward-weight)
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
  one "tree (tn512 wg384)" ""
  one "tn256             " tn256
  one "tn128             " tn128
  one "tn1024            " tn1024
  one "wg192             " wg192
  one "wg768             " wg768
done
