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
    # high-entropy magnitudes (the plain Irwin-Hall normal has < 8e5 distinct values: every threshold would sit in a
    # long run of ties)
    acc = ops.fill_normal(n, 3, 0, 1e-3) * (1.0 + ops.fill_uniform(n, 4, 0.0, 0.5))
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


def run_norm(iters):
    """Fused BatchNorm / GroupNorm kernels at the activation shapes of the benchmarks; bytes = algorithmic traffic
    per activation element (fp32 tensors read/written once per pass that needs them)."""
    res = {}

    def rec(name, sec, nbytes):
        gbs = nbytes / sec / 1e9
        res[name] = {"us": sec * 1e6, "alg_bytes": nbytes, "GBps": gbs, "frac_of_8TBps": gbs / HBM_PEAK_GBS}
        print(f"  {name:44s} {sec*1e6:10.1f} us  {gbs:8.1f} GB/s  ({gbs/HBM_PEAK_GBS:.3f} of 8 TB/s)", flush=True)

    for tag, shape in (("resnet l1 256x64x32x32", (256, 64, 32, 32)), ("resnet l4 256x512x4x4", (256, 512, 4, 4))):
        N, C, H, W = shape
        x = torch.randn(shape, device="cuda"); r = torch.randn(shape, device="cuda"); dy = torch.randn(shape, device="cuda")
        g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
        e = x.numel() * 4
        y, m, i = ops.bn_forward(x, r, g, b, rm, rv, True, 0.1, 1e-5, True)
        # forward: stats read x; apply reads x, res, writes y
        rec(f"bn_forward+res+relu {tag} (16B)", timeit(lambda: ops.bn_forward(x, r, g, b, rm, rv, True, 0.1, 1e-5, True), iters), 4 * e)
        # backward: reduce reads dy, y, x; apply reads dy, y, x, writes dx, dres
        rec(f"bn_backward+dres {tag} (32B)", timeit(lambda: ops.bn_backward(dy, y, x, g, m, i, True, True, True), iters), 8 * e)
    for tag, shape in (("ddpm 128x128x32x32", (128, 128, 32, 32)), ("ddpm 128x256x16x16", (128, 256, 16, 16)),
                       ("sd 8x320x64x64", (8, 320, 64, 64))):
        N, C, H, W = shape
        x = torch.randn(shape, device="cuda"); dz = torch.randn(shape, device="cuda")
        g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        e = x.numel() * 4
        z, m, rs = ops.gn_forward(x, g, b, 32, 1e-6, True)
        rec(f"gn_forward+silu {tag} (8B)", timeit(lambda: ops.gn_forward(x, g, b, 32, 1e-6, True), iters), 2 * e)
        rec(f"gn_backward+silu {tag} (12B)", timeit(lambda: ops.gn_backward(dz, x, g, b, m, rs, 32, True), iters), 3 * e)
    return res


def run_next(n, iters):
    """K9 / K10 on flat vectors of n elements."""
    res = {}
    p0 = ops.fill_normal(n, 1, 0, 0.05)
    p = p0 + ops.fill_normal(n, 2, 0, 0.01)
    F = ops.fill_uniform(n, 3, 0.0, 50.0)
    g = ops.fill_normal(n, 4, 0, 1e-3)
    scratch = torch.empty_like(p); sm = torch.empty(n, dtype=torch.uint8, device="cuda")

    def rec(name, sec, bpe):
        gbs = bpe * n / sec / 1e9
        res[name] = {"us": sec * 1e6, "alg_bytes_per_elem": bpe, "GBps": gbs, "frac_of_8TBps": gbs / HBM_PEAK_GBS}
        print(f"  {name:44s} {sec*1e6:10.1f} us  {gbs:8.1f} GB/s  ({gbs/HBM_PEAK_GBS:.3f} of 8 TB/s)", flush=True)

    rec("ewc_penalty_grad(20B)", timeit(lambda: ops.ewc_penalty_grad(p, p0, F, g, 10.0), iters), 20)
    # each call starts from the same fresh p (a re-applied step would rank a vector whose lower quarter already ties at 0,
    # i.e. the top-k's full-scan fallback, not the step the unlearning loop takes): copy outside the timed region
    q = p.clone()
    tot, reps = 0.0, max(iters // 5, 3)
    for i in range(reps + 1):
        q.copy_(p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.proximal_step(q, p0, n // 4, scratch, sm)
        e1.record()
        torch.cuda.synchronize()
        if i:
            tot += e0.elapsed_time(e1) * 1e-3
    rec("proximal_step(diff+select+soft, 24B)", tot / reps, 24)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="n18,nd")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--json", default=None)
    ap.add_argument("--extra", action="store_true", help="also time the fused norm kernels and K9/K10")
    a = ap.parse_args()
    out = {}
    for s in a.sizes.split(","):
        n = SIZES[s]
        print(f"== {s}: N = {n:,}", flush=True)
        out[s] = run(n, a.iters if s != "ns" else max(a.iters // 5, 3))
        torch.cuda.empty_cache()
    if a.extra:
        print("== fused norm kernels", flush=True)
        out["norm"] = run_norm(a.iters)
        print(f"== next rows at N18", flush=True)
        out["next_n18"] = run_next(SIZES["n18"], a.iters)
        print(f"== next rows at N_D", flush=True)
        out["next_nd"] = run_next(SIZES["nd"], a.iters)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
