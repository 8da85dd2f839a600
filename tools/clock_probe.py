"""Which clock does the chip sustain under the unlearning step?  A one-wave probe kernel (salun_clock_probe) on a stream
of its own samples shader cycles against the 100 MHz constant counter while (a) nothing else runs, (b) one convolution
kernel loops, (c) the ResNet-18 unlearning step of bench.py loops.  Prints MHz per phase (median / min / max of samples)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn as nn
from unlearn_saliency_amd import _lib, ops

dev = torch.device("cuda")
probe_stream = torch.cuda.Stream()
L = _lib.lib()


_bufs = [torch.zeros(2 * 64, dtype=torch.int64, device=dev) for _ in range(8)]
torch.cuda.synchronize()   # the probe stream does not wait for the main stream: its buffers exist before any work is queued


def probe(samples=40, spins=200):
    out = _bufs.pop()[:2 * samples]
    with torch.cuda.stream(probe_stream):
        _lib.check(L.salun_clock_probe(ctypes.c_void_p(out.data_ptr()), samples, spins,
                                       ctypes.c_void_p(probe_stream.cuda_stream)), "salun_clock_probe")
    return out


def mhz(out):
    a = out.cpu().numpy().astype(np.float64).reshape(-1, 2)
    a = a[a[:, 1] > 0]
    f = a[:, 0] / a[:, 1] * 100.0
    return f"{np.median(f):7.1f} MHz (min {f.min():7.1f}, max {f.max():7.1f}; {len(f)} samples of {a[:,1].mean() / 100:.0f} us)"


torch.cuda.synchronize()
o = probe(); torch.cuda.synchronize(); print("idle chip:                 ", mhz(o))
# (b) one kernel
N, C, H, K = 256, 128, 16, 128
x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05; dy = torch.randn(N, K, H, H, device=dev)
imd = ops.conv3x3_pack(w, True)
for _ in range(200):
    ops.conv3x3_packed(dy, imd, C)
o = probe()
for _ in range(400):
    ops.conv3x3_packed(dy, imd, C)
torch.cuda.synchronize(); print("ring backward-data loop:   ", mhz(o))
for _ in range(100):
    ops.conv2d_backward_weight(x, dy, w.shape, 1, 1)
o = probe()
for _ in range(300):
    ops.conv2d_backward_weight(x, dy, w.shape, 1, 1)
torch.cuda.synchronize(); print("ring backward-weight loop: ", mhz(o))
s2 = torch.cuda.Stream()
def pair():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        ops.conv2d_backward_weight(x, dy, w.shape, 1, 1, shared=True)
    ops.conv3x3_packed(dy, imd, C)
    torch.cuda.current_stream().wait_stream(s2)
for _ in range(60):
    pair()
o = probe()
for _ in range(200):
    pair()
torch.cuda.synchronize(); print("backward-data || weight:   ", mhz(o))
# (c) the step
import bench
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    model, fl, rl = bench.build_workload(dev, 0, 1, 256)
from unlearn_saliency_amd.conv import use_salun_convs
from unlearn_saliency_amd.norm import use_fused_bn
from unlearn_saliency_amd.flat import arena_of
from unlearn_saliency_amd.Classification.unlearn.impl import FusedMaskedSGD
use_salun_convs(model); use_fused_bn(model)
arena = arena_of(model)
opt = FusedMaskedSGD(arena, 0.013, momentum=0.9, weight_decay=5e-4)
opt.set_mask(ops.mask_topk(ops.fill_normal(arena.n, 5, 0.0, 1e-3), [arena.n // 2])[0])
model.train()
crit = nn.CrossEntropyLoss()
stream = iter(bench.StepStream(fl, rl))
def step():
    xb, yb, _ = next(stream)
    loss = crit(model(xb), yb)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(15):
    step()
o = probe(samples=60, spins=200)
for _ in range(60):
    step()
torch.cuda.synchronize(); print("ResNet-18 unlearning steps:", mhz(o))
