"""Fused attention (K13) against the library's scaled_dot_product_attention at the SD-v1 U-Net shapes (batch 8, 8 heads).
python tools/attnbench.py [--iters 10]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from unlearn_saliency_amd import ops

SHAPES = [(4096, 4096, 40), (4096, 77, 40), (1024, 1024, 80), (1024, 77, 80), (256, 256, 160), (256, 77, 160), (64, 64, 160), (64, 77, 160)]


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    B, H = a.batch, 8
    print(f"{'Nq x Nk, D':22s} {'GFLOP fwd':>9s} | {'fwd us':>8s} {'TF':>6s} {'lib us':>8s} | {'bwd us':>8s} {'TF':>6s} {'lib us':>8s}")
    tf = tb = lf = lb = 0.0
    for Nq, Nk, D in SHAPES:
        q = torch.randn(B, Nq, H * D, device="cuda").to(torch.bfloat16).requires_grad_(True)
        k = torch.randn(B, Nk, H * D, device="cuda").to(torch.bfloat16).requires_grad_(True)
        v = torch.randn(B, Nk, H * D, device="cuda").to(torch.bfloat16).requires_grad_(True)
        view = lambda t: t.view(B, t.shape[1], H, D)
        gf = 4.0 * B * H * Nq * Nk * D / 1e9
        o = ops.attention(view(q), view(k), view(v), D ** -0.5)
        d_o = torch.randn_like(o)
        t_f = timeit(lambda: ops.attn_forward(view(q), view(k), view(v), D ** -0.5), a.iters)
        t_b = timeit(lambda: torch.autograd.grad(o, [q, k, v], d_o, retain_graph=True), a.iters)
        sp = lambda t: view(t).transpose(1, 2)
        ol = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=D ** -0.5)
        dl = torch.randn_like(ol)
        l_f = timeit(lambda: F.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=D ** -0.5), a.iters)
        l_b = timeit(lambda: torch.autograd.grad(ol, [q, k, v], dl, retain_graph=True), a.iters)
        print(f"{Nq:5d} x {Nk:5d}, D={D:<4d} {gf:9.1f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} {l_f:8.1f} | {t_b:8.1f} {2.5 * gf / t_b * 1e3:6.1f} {l_b:8.1f}", flush=True)
        tf += t_f; tb += t_b; lf += l_f; lb += l_b
    print(f"sum: fwd {tf:.0f} us (library {lf:.0f}), bwd {tb:.0f} us (library {lb:.0f})")


if __name__ == "__main__":
    main()
