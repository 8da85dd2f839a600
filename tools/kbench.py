"""Kernel micro-benchmark: achieved algorithmic GB/s of every HBM-bound kernel at the three
model sizes of SURVEY.md §8 (N18, N_D, optionally N_S).  Timing = HIP events on the
stream the kernels are launched on (torch's current stream), averaged over `iters` launches.

    python tools/kbench.py [--sizes n18,nd,ns] [--iters 50] [--json out.json]
"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from unlearn_saliency_amd import ops

SIZES = {"n18": 11_173_962, "nd": 38_632_323, "ns": 859_520_964}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters  # seconds per launch


def run(n, iters, nk_list=(1, 10)):
    res = {}
    p = ops.fill_normal(n, 1, 0, 0.05)
    g = ops.fill_normal(n, 2, 0, 1e-3)
    buf = torch.zeros(n, device="cuda")
    acc = ops.fill_normal(n, 3, 0, 1e-3)
    m = ops.mask_topk(acc, [n // 2])[0]

    def rec(name, sec, bytes_per_elem):
        gbs = bytes_per_elem * n / sec / 1e9
        res[name] = {"us": sec * 1e6, "alg_bytes_per_elem": bytes_per_elem, "GBps": gbs, "frac_of_8TBps": gbs / HBM_PEAK_GBS}
        print(f"  {name:28s} {sec*1e6:10.1f} us  {gbs:8.1f} GB/s  ({gbs/HBM_PEAK_GBS:.3f} of 8 TB/s)", flush=True)

    rec("masked_sgd(21B)", timeit(lambda: ops.masked_sgd_step(p, g, buf, m, 0.013, 0.9, 5e-4, False), iters), 21)
    rec("sgd_unmasked(20B)", timeit(lambda: ops.masked_sgd_step(p, g, buf, None, 0.013, 0.9, 5e-4, False), iters), 20)
    m1 = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    sq = ops.grad_sqnorm(g)
    rec("grad_sqnorm(4B)", timeit(lambda: ops.grad_sqnorm(g, sq), iters), 4)
    step = [0]

    def adam():
        step[0] += 1
        ops.masked_adam_step(p, g, m1, v, m, 1e-4, 0.9, 0.999, 1e-8, 0.0, step[0], sqnorm=sq, max_norm=1.0)

    rec("masked_adam(29B)", timeit(adam, iters), 29)
    rec("saliency_accumulate(12B)", timeit(lambda: ops.saliency_accumulate(acc, g, 1.0), iters), 12)
    del m1, v
    tmp = torch.zeros(n, device="cuda")
    rec("fim_square_accumulate(16B)", timeit(lambda: ops.fim_square_accumulate(buf, tmp, 5000.0), iters), 16)
    del tmp
    for nk in nk_list:
        ks = [int(n * (i + 1) / 10) for i in range(nk)] if nk > 1 else [n // 2]
        outs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in ks]
        rec(f"mask_topk nk={nk}({4+nk}B)", timeit(lambda: ops.mask_topk(acc, ks, outs), max(iters // 5, 3)), 4 + nk)
        del outs
    rec("mask_u8_to_i64(9B)", timeit(lambda: ops.mask_u8_to_i64(m), max(iters // 5, 3)), 9)
    # reference points: a plain device copy (8 B/elem) through torch
    dst = torch.empty_like(p)
    rec("torch_copy(8B)", timeit(lambda: dst.copy_(p), iters), 8)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="n18,nd")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    out = {}
    for s in a.sizes.split(","):
        n = SIZES[s]
        print(f"== {s}: N = {n:,}", flush=True)
        out[s] = run(n, a.iters if s != "ns" else max(a.iters // 5, 3))
        torch.cuda.empty_cache()
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
