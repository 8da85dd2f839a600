"""Per-layer conv micro-benchmark: MFMA kernels (salun_conv2d_*) vs the library path (F.conv2d / autograd),
ResNet-18 CIFAR shapes at batch 256 (and DDPM shapes with --ddpm).  Reports ms and TFLOP/s (direct-conv FLOPs).
The weights are registered with ringpack (as conv.use_salun_convs does for a model), so 3x3 / stride 1 layers take the
kernels the models take: the LDS-DMA ring kernel forward and backward-data; backward-weight is timed both ways — alone
(ring kernel) and flagged SALUN_WGRAD_SHARED (conv_wgrad_v, what the side stream of the training step runs).
Timed at the SUSTAINED clock: 120 warm-up launches, 60 timed (tools/convring_bench.py has the reason)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from unlearn_saliency_amd import ops, ringpack

RESNET = [  # (name, N, C, H, K, R, stride, pad, count per step)
    ("stem 3->64 @32", 256, 3, 32, 64, 3, 1, 1, 1),
    ("l1 64->64 @32", 256, 64, 32, 64, 3, 1, 1, 4),
    ("l2 64->128 s2", 256, 64, 32, 128, 3, 2, 1, 1),
    ("l2 ds 1x1 s2", 256, 64, 32, 128, 1, 2, 0, 1),
    ("l2 128->128 @16", 256, 128, 16, 128, 3, 1, 1, 3),
    ("l3 128->256 s2", 256, 128, 16, 256, 3, 2, 1, 1),
    ("l3 ds 1x1 s2", 256, 128, 16, 256, 1, 2, 0, 1),
    ("l3 256->256 @8", 256, 256, 8, 256, 3, 1, 1, 3),
    ("l4 256->512 s2", 256, 256, 8, 512, 3, 2, 1, 1),
    ("l4 ds 1x1 s2", 256, 256, 8, 512, 1, 2, 0, 1),
    ("l4 512->512 @4", 256, 512, 4, 512, 3, 1, 1, 3),
]
DDPM = [
    ("conv_in 3->128 @32", 128, 3, 32, 128, 3, 1, 1, 1),
    ("128->128 @32", 128, 128, 32, 128, 3, 1, 1, 6),
    ("256->256 @16", 128, 256, 16, 256, 3, 1, 1, 8),
    ("512->256 @16", 128, 512, 16, 256, 3, 1, 1, 3),
    ("256->256 @8", 128, 256, 8, 256, 3, 1, 1, 8),
    ("256->256 @4", 128, 256, 4, 256, 3, 1, 1, 8),
    ("1x1 256->256 @16", 128, 256, 16, 256, 1, 1, 0, 20),
]


def timeit(fn, iters=60, warm=120):  # sustained clock: see tools/convring_bench.py
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ddpm", action="store_true")
    ap.add_argument("--no_lib", action="store_true")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "lib_fwd": 0.0, "lib_bwd": 0.0}
    tot["wgrad_shared"] = 0.0
    keep = []
    print(f"{'layer':22s} {'GF':>7s} | {'fwd ms':>8s} {'TF':>6s} | {'dgrad':>8s} {'TF':>6s} | {'wgrad':>8s} {'TF':>6s} | {'w shared':>8s} {'TF':>6s} | lib fwd / bwd(dx+dw) ms")
    for name, N, C, H, K, R, s, p, cnt in (DDPM if a.ddpm else RESNET):
        P = (H + 2 * p - R) // s + 1
        x = torch.randn(N, C, H, H, device="cuda")
        w = torch.randn(K, C, R, R, device="cuda") * 0.05
        keep.append(w)
        ringpack.register([w])  # (3x3 / stride 1 / C, K multiples of 8 only: the others are refused there)
        dy = torch.randn(N, K, P, P, device="cuda")
        gf = 2.0 * N * K * P * P * C * R * R / 1e9
        t_f = timeit(lambda: ops.conv2d_forward(x, w, None, s, p, P, P))
        t_d = timeit(lambda: ops.conv2d_backward_data(dy, w, x.shape, s, p)) if C > 3 else 0.0
        t_w = timeit(lambda: ops.conv2d_backward_weight(x, dy, w.shape, s, p))
        t_ws = timeit(lambda: ops.conv2d_backward_weight(x, dy, w.shape, s, p, shared=True))
        lf = lb = float("nan")
        if not a.no_lib:
            lf = timeit(lambda: F.conv2d(x, w, None, s, p), iters=5, warm=2)
            xr, wr = x.clone().requires_grad_(C > 3), w.clone().requires_grad_(True)
            def bwd():
                y = F.conv2d(xr, wr, None, s, p)
                torch.autograd.grad(y, [t for t in (xr, wr) if t.requires_grad], dy)
            lb = timeit(bwd, iters=5, warm=2) - lf
        tf = lambda ms: gf / ms if ms > 0 else 0.0
        print(f"{name:22s} {gf:7.1f} | {t_f:8.3f} {tf(t_f):6.1f} | {t_d:8.3f} {tf(t_d):6.1f} | {t_w:8.3f} {tf(t_w):6.1f} | {t_ws:8.3f} {tf(t_ws):6.1f} | {lf:8.3f} / {lb:8.3f}   x{cnt}", flush=True)
        tot["fwd"] += cnt * t_f; tot["dgrad"] += cnt * t_d; tot["wgrad"] += cnt * t_w; tot["wgrad_shared"] += cnt * t_ws
        tot["lib_fwd"] += cnt * lf; tot["lib_bwd"] += cnt * lb
    print("per-step conv totals (ms):", {k: round(v, 3) for k, v in tot.items()},
          " salun sum:", round(tot["fwd"] + tot["dgrad"] + tot["wgrad"], 3))


if __name__ == "__main__":
    main()
