"""salun_mask_topk (one threshold) at several vector sizes: sampled single-pass path vs the full-scan path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for n in (1 << 26, 1 << 27, 1 << 28, 859_520_964):
    d = ops.fill_normal(n, 123, 0.0, 1e-3) * (1.0 + ops.fill_uniform(n, 124, 0.0, 0.5))
    out = [torch.empty(n, dtype=torch.uint8, device="cuda")]
    k = [n // 2]
    os.environ["SALUN_TOPK_SAMPLED_MIN"] = "1"
    ts = timed(lambda: ops.mask_topk(d, k, out))
    ms = out[0].clone()
    os.environ["SALUN_TOPK_SAMPLED_MIN"] = str(1 << 40)
    tf = timed(lambda: ops.mask_topk(d, k, out))
    same = torch.equal(ms, out[0])
    del os.environ["SALUN_TOPK_SAMPLED_MIN"]
    print(f"N = {n:>11,d}: sampled {ts:7.3f} ms ({5*n/ts/1e6:7.1f} GB/s)   full scan {tf:7.3f} ms ({5*n/tf/1e6:7.1f} GB/s)   same mask: {same}", flush=True)
    del d, out, ms
    torch.cuda.empty_cache()
