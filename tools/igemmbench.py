"""Forward / backward-data only, three ResNet-18 shapes (batch 256), 30 calls each; with SALUN_LIB=<A/B build>
(csrc/salun_conv.hip: SALUN_IGEMM_EXP) it shows where conv_igemm's time goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

SHAPES = [("l1 64->64 @32", 256, 64, 32, 64, 3, 1, 1), ("l2 128->128 @16", 256, 128, 16, 128, 3, 1, 1),
          ("l4 512->512 @4", 256, 512, 4, 512, 3, 1, 1)]


def t(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30


def main():
    tag = os.path.basename(os.environ.get("SALUN_LIB", "product"))
    for name, N, C, H, K, R, s, p in SHAPES:
        P = (H + 2 * p - R) // s + 1
        x = torch.randn(N, C, H, H, device="cuda")
        w = torch.randn(K, C, R, R, device="cuda") * 0.05
        dy = torch.randn(N, K, P, P, device="cuda")
        gf = 2.0 * N * K * P * P * C * R * R / 1e9
        f = t(lambda: ops.conv2d_forward(x, w, None, s, p, P, P))
        d = t(lambda: ops.conv2d_backward_data(dy, w, x.shape, s, p))
        print(f"{tag:22s} {name:16s} fwd {f * 1e3:7.1f} us {gf / f:6.1f} TF | dgrad {d * 1e3:7.1f} us {gf / d:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
