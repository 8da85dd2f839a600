cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ddpm_block_gpu.py tests/test_norm_gpu.py -q -m gpu -x --timeout 600 -s 2>&1 | tail -25 > gpurun_out/blk_tests.txt; cat gpurun_out/blk_tests.txt
timeout 900 python -m pytest tests/test_ddpm_gpu.py tests/test_conv_gpu.py tests/test_f4_gpu.py -q -m gpu --timeout 600 2>&1 | tail -12 > gpurun_out/blk_tests2.txt; cat gpurun_out/blk_tests2.txt
timeout 300 python tools/kbench.py 2>/dev/null | grep -E "gn_|bn_" > gpurun_out/blk_kbench.txt; cat gpurun_out/blk_kbench.txt
for nodes in 0 1; do
  SALUN_BLOCK_NODES=$nodes timeout 600 python tools/bench_ddpm.py --no_cpu_baseline --mask_batches 2 --steps 20 --warmup 3 > gpurun_out/blk_ddpm_$nodes.json 2>gpurun_out/blk_ddpm_$nodes.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/blk_ddpm_$nodes.json').read().strip().splitlines()[-1]); print('ddpm nodes=$nodes', d['value'], d['ms_per_step'], d.get('fwd_bwd',{}).get('frac'), d.get('library_conv_calls'))" || tail -5 gpurun_out/blk_ddpm_$nodes.err
done
python bench.py --steps 177 --warmup 10 --no_cpu_baseline > gpurun_out/blk_bench.json 2> gpurun_out/blk_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/blk_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['fwd_bwd']['frac'])"
