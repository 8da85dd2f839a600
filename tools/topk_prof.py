"""salun_mask_topk at one (size, nk) configuration, a few launches — the command rocprofv3 wraps for the
per-kernel breakdown (tools/prof.sh) and the PMC passes (tools/pmc.sh).

    python tools/topk_prof.py n18|nd|ns|<N> <nk> [iters] [flags]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

SIZES = {"n18": 11_173_962, "nd": 38_632_323, "ns": 859_520_964}
n = SIZES.get(sys.argv[1]) or int(sys.argv[1])
nk = int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
acc = ops.fill_normal(n, 3, 0.0, 1e-3) * (1.0 + ops.fill_uniform(n, 4, 0.0, 0.5))  # few ties (fill_normal alone has 786 K values)
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ks = [int(n * (i + 1) / 10) for i in range(nk)] if nk > 1 else [n // 2]
outs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in ks]
for _ in range(2):
    ops.mask_topk(acc, ks, outs, flags=flags)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    ops.mask_topk(acc, ks, outs, flags=flags)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / iters
print(f"mask_topk n={n} nk={nk}: {us:.1f} us/launch, {(4 + nk) * n / us / 1e3:.1f} GB/s algorithmic "
      f"({(4 + nk) * n / us / 1e3 / 8000:.3f} of 8 TB/s); popcounts ok: "
      f"{all(ops.mask_popcount(o) == k for o, k in zip(outs, ks))}; (route, error) = {ops.mask_topk_status(acc.device)}")
