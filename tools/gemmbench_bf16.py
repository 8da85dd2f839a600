"""K16 bf16 GEMM on the SD transformer shapes (batch 8, 64x64 latents): every tile variant vs the library GEMM
(torch.mm -> hipBLASLt) vs the K11 1x1 convolution route.   python tools/gemmbench_bf16.py [--reps 30]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3  # us


VARS = (3, 8, 9, 10, 11, 12, 13)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    from unlearn_saliency_amd import ops
    shapes = []
    for M, d in ((32768, 320), (8192, 640), (2048, 1280), (512, 1280)):
        shapes += [(M, d, d), (M, 8 * d, d), (M, d, 4 * d)]
    shapes += [(616, 320, 768), (616, 1280, 768)]
    print(f"{'M':>6} {'N':>6} {'K':>5} | {'lib us':>8} {'TF':>6} | " + " ".join(f"{'v%d us' % v:>8} {'TF':>6}" for v in VARS)
          + f" | {'K11 us':>8} {'TF':>6}")
    tot = {"lib": 0.0, "best": 0.0, "k11": 0.0}
    for M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda")
        fl = 2.0 * M * N * K
        t_lib = timeit(lambda: torch.nn.functional.linear(x, w, bias.bfloat16()), a.reps)
        row = f"{M:>6} {N:>6} {K:>5} | {t_lib:8.1f} {fl / t_lib / 1e6:6.0f} | "
        best = 1e9
        for v in VARS:
            if v in (1, 2, 5, 7, 8, 10, 11, 12) and N % 128:
                row += f"{'-':>8} {'-':>6} "
                continue
            t = timeit(lambda: ops.gemm_bf16_nt(x, w, bias, None, v), a.reps)
            best = min(best, t)
            row += f"{t:8.1f} {fl / t / 1e6:6.0f} "
        wp = w.view(N, 1, K)
        xn = x.view(1, M // 8, 8, K)
        t11 = timeit(lambda: ops.conv2d_bf16_forward(xn, wp, 1, 1, 0, bias=bias), a.reps)
        row += f"| {t11:8.1f} {fl / t11 / 1e6:6.0f}"
        print(row, flush=True)
        tot["lib"] += t_lib; tot["best"] += best; tot["k11"] += t11
    print(f"sum over shapes: library {tot['lib']:.0f} us, best K16 variant {tot['best']:.0f} us, K11 {tot['k11']:.0f} us")
    tn(a, shapes)


def tn(a, shapes):
    """Weight gradients dW[N, K] = dY[M, N]^T . X[M, K]: K16's TN variants vs the library (bf16 result) vs K11's tap kernel."""
    from unlearn_saliency_amd import ops
    os.environ["SALUN_WGRAD_TN"] = "0"   # read once by the library: the K11 column below is the tap kernel
    print(f"\nweight gradient  {'M':>6} {'N':>6} {'K':>5} | {'lib us':>8} {'TF':>6} | "
          + " ".join(f"{'tn%d us' % v:>8} {'TF':>6}" for v in (1, 2, 3)) + f" | {'K11 us':>8} {'TF':>6}")
    tot = {"lib": 0.0, "best": 0.0, "k11": 0.0}
    for M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").bfloat16()
        dy = torch.randn(M, N, device="cuda").bfloat16()
        dw = torch.zeros(N, K, device="cuda")
        fl = 2.0 * M * N * K
        t_lib = timeit(lambda: torch.mm(dy.t(), x), a.reps)
        row = f"                 {M:>6} {N:>6} {K:>5} | {t_lib:8.1f} {fl / t_lib / 1e6:6.0f} | "
        best = 1e9
        for v in (1, 2, 3):
            t = timeit(lambda: ops.gemm_bf16_tn(dy, x, out=dw, accumulate=True, variant=v), a.reps)
            best = min(best, t)
            row += f"{t:8.1f} {fl / t / 1e6:6.0f} "
        if M % 8 == 0:
            xn, dyn = x.view(1, M // 8, 8, K), dy.view(1, M // 8, 8, N)
            t11 = timeit(lambda: ops.conv2d_bf16_backward_weight(xn, dyn, (N, K, 1, 1), 1, 0, out=dw.view(N, K, 1, 1), accumulate=True), a.reps)
        else:
            t11 = float("nan")
        row += f"| {t11:8.1f} {fl / t11 / 1e6:6.0f}"
        print(row, flush=True)
        tot["lib"] += t_lib; tot["best"] += best; tot["k11"] += t11
    print(f"sum over shapes: library {tot['lib']:.0f} us, best TN variant {tot['best']:.0f} us, K11 tap kernel {tot['k11']:.0f} us")


if __name__ == "__main__":
    main()
