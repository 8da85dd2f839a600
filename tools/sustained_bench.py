"""Does a convolution kernel hold its burst rate?  Times blocks of 100 back-to-back launches for ~0.5 s per kernel
(burst measurements of tools/convbench.py last 3 ms) and samples rocm-smi's clock / power read-out meanwhile.
Also the same two kernels (backward-data + backward-weight of one layer) on two streams at once."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

N, C, H, K = 256, 128, 16, 128
x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05
dy = torch.randn(N, K, H, H, device="cuda")
imf, imd = ops.conv3x3_pack(w, False), ops.conv3x3_pack(w, True)
gf = 2.0 * N * K * H * H * C * 9 / 1e9
smi = []
stop = [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [ln.strip() for ln in out.splitlines() if "sclk" in ln or "Power" in ln or "mclk" in ln]
            smi.append((time.time(), " | ".join(s)[:200]))
        except Exception as e:
            smi.append((time.time(), repr(e)[:80]))
        time.sleep(0.1)


def blocks(fn, nblk=30, per=100, streams=None):
    res = []
    torch.cuda.synchronize()
    for b in range(nblk):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per):
            fn()
        e1.record()
        res.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) / per * 1e3 for a, b in res]


th = threading.Thread(target=sampler, daemon=True); th.start()
for name, fn, g in (("ring dgrad", lambda: ops.conv3x3_packed(dy, imd, C), gf),
                    ("igemm dgrad", lambda: ops.conv2d_backward_data(dy, w, x.shape, 1, 1), gf),
                    ("wgrad", lambda: ops.conv2d_backward_weight(x, dy, w.shape, 1, 1), gf)):
    t0 = time.time()
    us = blocks(fn)
    print(f"{name:12s} us per call by block of 100: " + " ".join(f"{u:.0f}" for u in us) + f"  | TF first {g / us[0] * 1e3:.1f} last {g / us[-1] * 1e3:.1f}", flush=True)
    print("   smi:", [s for t, s in smi if t >= t0][-1:] )
# the pair on two streams
s2 = torch.cuda.Stream()
def pair():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        ops.conv2d_backward_weight(x, dy, w.shape, 1, 1)
    ops.conv3x3_packed(dy, imd, C)
    torch.cuda.current_stream().wait_stream(s2)
us = blocks(pair, nblk=20, per=50)
print("pair ring-dgrad || wgrad: us per pair " + " ".join(f"{u:.0f}" for u in us) + f" | TF {2 * gf / us[-1] * 1e3:.1f}")
def pair2():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        ops.conv2d_backward_weight(x, dy, w.shape, 1, 1)
    ops.conv2d_backward_data(dy, w, x.shape, 1, 1)
    torch.cuda.current_stream().wait_stream(s2)
us = blocks(pair2, nblk=20, per=50)
print("pair igemm-dgrad || wgrad: us per pair " + " ".join(f"{u:.0f}" for u in us) + f" | TF {2 * gf / us[-1] * 1e3:.1f}")
stop[0] = True
print("smi samples:", len(smi)); print("\n".join(s for _, s in smi[::5][:12]))
