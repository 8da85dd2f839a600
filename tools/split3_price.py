"""Price of an fp32-accurate product on the bf16 matrix cores by 3-way operand split (last part).

a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1) (8 + 8 + 8 mantissa bits), same for b;
a.b ~= a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0): six bf16 products, fp32 accumulation (smallest terms first).
Measured on the one kernel of this repository that multiplies bf16 operands INTO an fp32 result with accumulation — K16's
weight-gradient GEMM dW += dY^T . X — against K15 (v_mfma_f32_32x32x2_f32, exact fp32 products) on the same operands,
errors against float64.  The splitting passes (element-wise, 3 pieces per operand) are timed too.
   python tools/split3_price.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import gemm, ops


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


def split3(a):
    a0 = a.bfloat16()
    r = a - a0.float()
    a1 = r.bfloat16()
    a2 = (r - a1.float()).bfloat16()
    return a0, a1, a2


def main():
    print(f"{'M (reduction)':>14} {'Na':>5} {'Nb':>5} | {'K15 fp32 us':>11} {'TF':>5} {'err':>8} | {'1 x bf16 us':>11} {'TF':>5} {'err':>8} | "
          f"{'6 x bf16 us':>11} {'TF-eq':>5} {'err':>8} | {'split us':>8}")
    for M, Na, Nb in ((16384, 1024, 1024), (131072, 128, 1152), (8192, 1280, 1280)):
        a = torch.randn(M, Na, device="cuda")
        b = torch.randn(M, Nb, device="cuda")
        want = a.double().t() @ b.double()
        scale = float(want.abs().max())
        fl = 2.0 * M * Na * Nb
        at = a.t()  # K15 reads any strides: dW[Na, Nb] = a^T . b as mm_nt(a^T [Na, M], b^T [Nb, M])
        t15 = timeit(lambda: gemm.mm_nt(at, b.t()))
        e15 = float((gemm.mm_nt(at, b.t()).double() - want).abs().max()) / scale
        a3, b3 = split3(a), split3(b)
        out = torch.zeros(Na, Nb, device="cuda")
        t1 = timeit(lambda: ops.gemm_bf16_tn(a3[0], b3[0], out=out, accumulate=False))
        e1 = float((ops.gemm_bf16_tn(a3[0], b3[0]).double() - want).abs().max()) / scale
        pairs = ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))  # smallest terms first

        def six():
            for n, (i, j) in enumerate(pairs):
                ops.gemm_bf16_tn(a3[i], b3[j], out=out, accumulate=n > 0)
        t6 = timeit(six)
        six()
        e6 = float((out.double() - want).abs().max()) / scale
        ts = timeit(lambda: (split3(a), split3(b)))
        print(f"{M:>14} {Na:>5} {Nb:>5} | {t15:11.1f} {fl / t15 / 1e6:5.0f} {e15:8.1e} | {t1:11.1f} {fl / t1 / 1e6:5.0f} {e1:8.1e} | "
              f"{t6:11.1f} {fl / t6 / 1e6:5.0f} {e6:8.1e} | {ts:8.1f}", flush=True)


if __name__ == "__main__":
    main()
