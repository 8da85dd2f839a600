"""Linear layers of the SD transformer blocks (batch 8): the library path (bf16 autocast F.linear + autograd, incl. the
weight cast and the fp32 AccumulateGrad it implies) vs SalunLinearBF16 (K11 1x1 kernels, cached packed weights, fp32
gradient accumulation in the kernel).  us per forward / backward, TFLOP/s of 2*M*C*K per GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd.conv_bf16 import SalunLinearBF16

SHAPES = [("to_q 64x64", 32768, 320, 320, False), ("ff.proj 64x64", 32768, 320, 2560, True),
          ("ff.out 64x64", 32768, 1280, 320, True), ("to_q 32x32", 8192, 640, 640, False),
          ("ff.proj 32x32", 8192, 640, 5120, True), ("ff.out 32x32", 8192, 2560, 640, True),
          ("ff.proj 16x16", 2048, 1280, 10240, True), ("ff.out 16x16", 2048, 5120, 1280, True),
          ("to_k ctx", 616, 768, 320, False), ("to_k ctx 1280", 616, 768, 1280, False)]


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for name, M, C, K, bias in SHAPES:
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    lib = torch.nn.Linear(C, K, bias=bias).cuda()
    own = torch.nn.Linear(C, K, bias=bias).cuda()
    own.__class__ = SalunLinearBF16
    for p in list(lib.parameters()) + list(own.parameters()):
        p.grad = torch.zeros_like(p)
    gf = 2.0 * M * C * K / 1e9

    def run(mod, bwd):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod(x)
        if bwd:
            y.backward(dy)

    res = {}
    for tag, mod in (("lib", lib), ("own", own)):
        f = t(lambda: run(mod, False))
        fb = t(lambda: run(mod, True))
        res[tag] = (f, fb - f)
    print(f"{name:16s} M={M:6d} {C:5d}->{K:5d} {gf:7.1f} GF | lib fwd {res['lib'][0]:7.1f} us {gf/res['lib'][0]*1e3:6.0f} TF  "
          f"bwd {res['lib'][1]:7.1f} us {2*gf/res['lib'][1]*1e3:6.0f} TF | own fwd {res['own'][0]:7.1f} us "
          f"{gf/res['own'][0]*1e3:6.0f} TF  bwd {res['own'][1]:7.1f} us {2*gf/res['own'][1]*1e3:6.0f} TF", flush=True)
