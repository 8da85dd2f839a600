"""salun_mask_topk at several vector sizes and threshold counts: the single-read route (default) and the persistent
full scan (FORCE_FULL_SCAN); prints which route published."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

FULL = 1


def timed(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [11_173_962, 38_632_323, 1 << 27, 859_520_964]
for n in sizes:
    d = ops.fill_normal(n, 123, 0.0, 1e-3) * (1.0 + ops.fill_uniform(n, 124, 0.0, 0.5))
    for nk in ((1, 10) if n < (1 << 27) else (1,)):
        ks = [int(n * (i + 1) / 10) for i in range(nk)] if nk > 1 else [n // 2]
        out = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in ks]
        line = f"N = {n:>11,d} nk = {nk:2d}:"
        ref = None
        for name, flags in (("single-read", 0), ("full-scan", FULL)):
            us = timed(lambda: ops.mask_topk(d, ks, out, flags=flags), 10 if n < (1 << 27) else 4)
            route, err = ops.mask_topk_status(d.device)
            gbs = (4 + nk) * n / us / 1e3
            same = "" if ref is None else f" same={all(torch.equal(a, b) for a, b in zip(ref, out))}"
            if ref is None:
                ref = [o.clone() for o in out]
            line += f"  {name} {us:9.1f} us {gbs:7.1f} GB/s ({gbs / 8000:.3f}) route={route} err={err}{same};"
        print(line, flush=True)
        del out, ref
    del d
    torch.cuda.empty_cache()
