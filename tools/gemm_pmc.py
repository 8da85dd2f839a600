"""A few launches of the K16 GEMM variants on two SD shapes, for the counter passes (tools/pmc_multi.sh).
   python tools/gemm_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

for M, N, K in ((32768, 2560, 320), (8192, 5120, 640)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    for v in (8, 10):
        for _ in range(5):
            ops.gemm_bf16_nt(x, w, None, None, v)
    for v in (2, 3):
        for _ in range(5):
            ops.gemm_bf16_tn(dy, x, out=dw, accumulate=True, variant=v)
    for _ in range(5):
        torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
