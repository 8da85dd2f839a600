import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops
n = 11_173_962
p0 = ops.fill_normal(n, 1, 0, 0.05)
q = p0 + ops.fill_normal(n, 2, 0, 0.01)
for i in range(4):
    ops.proximal_step(q, p0, n // 4)
    torch.cuda.synchronize()
    print(i, ops.mask_topk_status(q.device), int((q == p0).sum()))
