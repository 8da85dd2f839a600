"""Tiny driver for PMC passes: 10 launches each of the update kernels and 5 one-threshold top-k calls at the given size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_173_962
p = ops.fill_normal(n, 1, 0, 0.05); g = ops.fill_normal(n, 2, 0, 1e-3); buf = torch.zeros(n, device="cuda")
acc = ops.fill_normal(n, 3, 0, 1e-3) * (1.0 + ops.fill_uniform(n, 4, 0.0, 0.5)); m = ops.mask_topk(acc, [n // 2])[0]
m1 = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); sq = ops.grad_sqnorm(g)
for i in range(10):
    ops.masked_sgd_step(p, g, buf, m, 0.013, 0.9, 5e-4, False)
    ops.masked_adam_step(p, g, m1, v, m, 1e-4, 0.9, 0.999, 1e-8, 0.0, i + 1, sqnorm=sq, max_norm=1.0)
    ops.saliency_accumulate(acc, g, 1.0)
    ops.grad_sqnorm(g, sq)
for i in range(5):
    ops.mask_topk(acc, [n // 2])
torch.cuda.synchronize()
