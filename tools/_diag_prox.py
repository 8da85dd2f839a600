import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops
n = 11_173_962
p0 = ops.fill_normal(n, 1, 0, 0.05)
p = p0 + ops.fill_normal(n, 2, 0, 0.01)
d = p - p0
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
k = n - n // 4 + 1
out = [torch.empty(n, dtype=torch.uint8, device="cuda")]
for name, fl in (("masks", 0), ("values_only", 2), ("full", 1), ("full+vo", 3)):
    us = t(lambda: ops.mask_topk(d, [k], out if not (fl & 2) else None, flags=fl))
    print(name, f"{us:.1f} us", "route/err", ops.mask_topk_status(d.device), "tau", ops.mask_topk_thresholds(d.device, 1).item())
print("distinct |d| near tau:", torch.unique(d.abs()).numel())
q = p.clone()
print("proximal_step", t(lambda: ops.proximal_step(q, p0, n // 4)), "us")
a = d.abs(); tau = ops.mask_topk_thresholds(d.device, 1).item(); print("ties at tau:", int((a == tau).sum()))
