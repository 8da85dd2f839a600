"""HBM-bound bf16 kernels of the SD configuration in isolation (K12 GroupNorm+SiLU, K14 LayerNorm / GEGLU): microseconds,
algorithmic GB/s and fraction of the 8 TB/s HBM peak.  Algorithmic bytes = every operand read once + every result
written once (bf16 = 2 B); K12 reads x twice (statistics, apply), so its ceiling against this count is 2/3 (fwd), 3/5 (bwd).
python tools/kbench_bf16.py [--iters 30] [--json out.json]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import norm, ops

PEAK = 8000.0


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = {}

    def rec(name, sec, nbytes):
        gbs = nbytes / sec / 1e9
        res[name] = {"us": sec * 1e6, "alg_bytes": nbytes, "GBps": gbs, "frac_of_8TBps": gbs / PEAK}
        print(f"  {name:52s} {sec * 1e6:9.1f} us {gbs:8.1f} GB/s ({gbs / PEAK:.3f} of 8 TB/s)", flush=True)

    for N, C, H in [(8, 320, 64), (8, 640, 32), (8, 1280, 16), (8, 2560, 8)]:
        x = torch.randn(N, C, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gn = torch.nn.GroupNorm(32, C).cuda()
        xn = x.permute(0, 2, 3, 1)
        y, mr, ab = ops.gn_bf16_forward(xn, gn.weight, gn.bias, 32, gn.eps, True)
        dy = torch.randn_like(y)
        gw, gb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        n = x.numel()
        rec(f"K12 gn+silu fwd {N}x{C}x{H}x{H} (4B)", timeit(lambda: ops.gn_bf16_forward(xn, gn.weight, gn.bias, 32, gn.eps, True), a.iters), 4 * n)
        rec(f"K12 gn+silu bwd {N}x{C}x{H}x{H} (6B)", timeit(lambda: ops.gn_bf16_backward(dy, xn, gn.weight, mr, ab, 32, True, gw, gb, True), a.iters), 6 * n)
    for rows, C in [(8 * 4096, 320), (8 * 1024, 640), (8 * 256, 1280)]:
        x = torch.randn(rows, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
        ln = torch.nn.LayerNorm(C).cuda()
        y = ops.layer_norm_bf16(x, ln)
        dy = torch.randn_like(y)
        n = x.numel()
        rec(f"K14 layer_norm fwd {rows}x{C} (4B)", timeit(lambda: ops.layer_norm_bf16(x.detach(), ln), a.iters), 4 * n)
        rec(f"K14 layer_norm fwd+bwd {rows}x{C} (10B)", timeit(lambda: torch.autograd.grad(ops.layer_norm_bf16(x, ln), [x, ln.weight, ln.bias], dy), a.iters), 10 * n)
    for rows, F in [(8 * 4096, 1280), (8 * 1024, 2560), (8 * 256, 5120)]:
        h = torch.randn(rows, 2 * F, device="cuda").to(torch.bfloat16).requires_grad_(True)
        o = ops.geglu_bf16(h)
        do = torch.randn_like(o)
        n = o.numel()
        rec(f"K14 geglu fwd {rows}x{F} (6B)", timeit(lambda: ops.geglu_bf16(h.detach()), a.iters), 6 * n)
        rec(f"K14 geglu fwd+bwd {rows}x{F} (16B)", timeit(lambda: torch.autograd.grad(ops.geglu_bf16(h), [h], do), a.iters), 16 * n)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
