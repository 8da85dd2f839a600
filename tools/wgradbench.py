"""Backward-weight only, three ResNet-18 shapes (batch 256): mean time of `ops.conv2d_backward_weight`
(conv_wgrad + conv_wgrad_reduce) over 30 calls.  Used with SALUN_LIB=<A/B build> (csrc/salun_conv.hip:
SALUN_WGRAD_EXP) to see where the kernel's time goes; prints one line per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

SHAPES = [("l1 64->64 @32", 256, 64, 32, 64, 3, 1, 1), ("l3 256->256 @8", 256, 256, 8, 256, 3, 1, 1),
          ("l2 64->128 s2", 256, 64, 32, 128, 3, 2, 1), ("l3 ds 1x1 s2", 256, 128, 16, 256, 1, 2, 0),
          ("ddpm 1x1 256 @16", 128, 256, 16, 256, 1, 1, 0), ("ddpm 128->128 @32", 128, 128, 32, 128, 3, 1, 1)]


def main():
    tag = os.path.basename(os.environ.get("SALUN_LIB", "product"))
    for name, N, C, H, K, R, s, p in SHAPES:
        P = (H + 2 * p - R) // s + 1
        x = torch.randn(N, C, H, H, device="cuda")
        dy = torch.randn(N, K, P, P, device="cuda")
        gf = 2.0 * N * K * P * P * C * R * R / 1e9
        for _ in range(5):
            ops.conv2d_backward_weight(x, dy, (K, C, R, R), s, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.conv2d_backward_weight(x, dy, (K, C, R, R), s, p)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(f"{tag:22s} {name:16s} {ms * 1e3:8.1f} us  {gf / ms:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
