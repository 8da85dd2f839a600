"""Per-layer timing of the bf16 MFMA convolutions (K11) at the SD-v1 U-Net shapes, batch 8, 64x64 latents.
python tools/convbench_bf16.py [--iters 20] [--lib]   (--lib also times the library's bf16 channels_last convolution)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

# (H, C, K, R, stride)
LAYERS = [(64, 320, 320, 3, 1), (64, 640, 320, 3, 1), (64, 960, 320, 3, 1), (64, 320, 320, 3, 2), (64, 320, 320, 1, 1),
          (32, 320, 640, 3, 1), (32, 640, 640, 3, 1), (32, 1280, 640, 3, 1), (32, 960, 640, 3, 1), (32, 640, 640, 3, 2),
          (16, 640, 1280, 3, 1), (16, 1280, 1280, 3, 1), (16, 2560, 1280, 3, 1), (16, 1920, 1280, 3, 1),
          (16, 1280, 1280, 3, 2), (8, 1280, 1280, 3, 1), (8, 2560, 1280, 3, 1), (16, 1280, 1280, 1, 1)]


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lib", action="store_true")
    a = ap.parse_args()
    N = a.batch
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print(f"{'layer':34s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}" + ("  | lib fwd/dgrad/wgrad us" if a.lib else ""))
    for H, C, K, R, st in LAYERS:
        pad = R // 2
        x = torch.randn(N, H, H, C, device="cuda").to(torch.bfloat16)
        w = torch.randn(K, C, R, R, device="cuda") / (C * R * R) ** 0.5
        wp = ops.conv2d_bf16_pack(w)
        y = ops.conv2d_bf16_forward(x, wp, R, st, pad)
        dy = torch.randn_like(y)
        dw = torch.zeros_like(w)
        gf = 2.0 * y.numel() * C * R * R / 1e9
        t_f = timeit(lambda: ops.conv2d_bf16_forward(x, wp, R, st, pad), a.iters)
        t_d = timeit(lambda: ops.conv2d_bf16_backward_data(dy, wp, tuple(x.shape), R, st, pad), a.iters)
        t_w = timeit(lambda: ops.conv2d_bf16_backward_weight(x, dy, tuple(w.shape), st, pad, out=dw, accumulate=True), a.iters)
        line = f"N{N} {H:2d}x{H:<2d} {C:4d}->{K:<4d} {R}x{R} s{st}        {gf:7.1f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} | {t_d:8.1f} {gf / t_d * 1e3:6.1f} | {t_w:8.1f} {gf / t_w * 1e3:6.1f}"
        for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[k][0] += gf; tot[k][1] += t
        if a.lib:
            xl = x.permute(0, 3, 1, 2).requires_grad_(True)  # channels_last view
            wl = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            yl = torch.nn.functional.conv2d(xl, wl, None, st, pad)
            dyl = dy.permute(0, 3, 1, 2)
            l_f = timeit(lambda: torch.nn.functional.conv2d(xl, wl, None, st, pad), a.iters)
            l_d = timeit(lambda: torch.autograd.grad(yl, xl, dyl, retain_graph=True), a.iters)
            l_w = timeit(lambda: torch.autograd.grad(yl, wl, dyl, retain_graph=True), a.iters)
            line += f"  | {l_f:8.1f} {l_d:8.1f} {l_w:8.1f}"
        print(line, flush=True)
    for k, (gf, t) in tot.items():
        print(f"total {k}: {gf:.0f} GFLOP in {t / 1e3:.2f} ms = {gf / t * 1e3:.1f} TFLOP/s ({gf / t * 1e3 / 2500:.3f} of 2.5 PF)")


if __name__ == "__main__":
    main()
