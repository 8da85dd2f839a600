"""What a memory-bound kernel pays for sharing the chip with a backward-weight kernel: BatchNorm backward (partial +
apply, layer-1 shape of ResNet-18 at batch 256) timed alone and while a second stream loops backward-weight launches.
Run under SALUN_LIB=<A/B build> to compare backward-weight kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

N, C, H = 256, 64, 32
x = torch.randn(N, C, H, H, device="cuda"); y = torch.relu(torch.randn(N, C, H, H, device="cuda"))
dy = torch.randn(N, C, H, H, device="cuda")
gamma = torch.randn(C, device="cuda"); mean = torch.randn(C, device="cuda"); invstd = torch.rand(C, device="cuda") + 0.5
s2 = torch.cuda.Stream()


def bn():
    ops.bn_backward(dy, y, x, gamma, mean, invstd, True, True, False)


def wg():
    ops.conv2d_backward_weight(x, dy, (C, C, 3, 3), 1, 1)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    return e0, e1


for _ in range(20):
    bn(); wg()
torch.cuda.synchronize()
a, b = timed(bn, 200); torch.cuda.synchronize()
print(f"BN backward alone: {a.elapsed_time(b) / 200 * 1e3:.1f} us")
a, b = timed(wg, 100); torch.cuda.synchronize()
print(f"backward-weight alone: {a.elapsed_time(b) / 100 * 1e3:.1f} us")
# co-run: stream 2 loops backward-weight for longer than the BN loop lasts
s2.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s2):
    c, d = timed(wg, 150)
a, b = timed(bn, 200)
torch.cuda.synchronize()
print(f"co-run: BN backward {a.elapsed_time(b) / 200 * 1e3:.1f} us per call (200 calls) beside backward-weight at {c.elapsed_time(d) / 150 * 1e3:.1f} us per call (150 calls)")
