"""K15 fp32 GEMM in the three orientations of a Linear layer (y = x.W^T, dx = dy.W, dW = dy^T.x) on SD / DDPM shapes
against the library (torch.mm -> hipBLASLt / rocBLAS fp32).   python tools/gemmbench_f32.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import gemm


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    print(f"{'M':>6} {'N':>6} {'K':>5} | " + " | ".join(f"{n + ' K15 us':>12} {'TF':>4} {'lib us':>8} {'TF':>4}" for n in ("y", "dx", "dW")))
    for M, N, K in ((32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640),
                    (2048, 1280, 1280), (2048, 10240, 1280), (616, 1280, 768), (128, 1024, 1024), (128, 256, 1024)):
        x, w, dy = (torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(M, N, device="cuda"))
        fl = 2.0 * M * N * K
        row = f"{M:>6} {N:>6} {K:>5} | "
        cells = []
        for own, lib in ((lambda: gemm.mm_nt(x, w), lambda: x @ w.t()),
                         (lambda: gemm.mm_nt(dy, w.t()), lambda: dy @ w),
                         (lambda: gemm.mm_nt(dy.t(), x.t()), lambda: dy.t() @ x)):
            t, tl = timeit(own), timeit(lib)
            cells.append(f"{t:12.1f} {fl / t / 1e6:4.0f} {tl:8.1f} {fl / tl / 1e6:4.0f}")
        print(row + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
