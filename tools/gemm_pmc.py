"""A few launches of the kernels added in round 4 on one fixed shape each, for the counter passes
(tools/pmc_multi.sh: SQ counters; tools/pmc.sh FETCH_SIZE / WRITE_SIZE: HBM traffic next to the algorithmic bytes).
   python tools/gemm_pmc.py
Shapes (algorithmic bytes per launch in tools/_run_r4_final.sh):
  K16 NT  y[32768, 2560] = x[32768, 320] . w[2560, 320]^T       bf16 in / out
  K16 TN  dw[2560, 320] += dy[32768, 2560]^T . x[32768, 320]    bf16 in, fp32 out (split partials + fold)
  K15     y[4096, 1024]  = x[4096, 1024] . w[1024, 1024]^T      fp32
  dropout 128 x 128 x 32 x 32 fp32, p = 0.1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import gemm, ops

M, N, K = 32768, 2560, 320
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
dy = torch.randn(M, N, device="cuda").bfloat16()
dw = torch.zeros(N, K, device="cuda")
for _ in range(6):
    ops.gemm_bf16_nt(x, w, None, None, 0)
for _ in range(6):
    ops.gemm_bf16_tn(dy, x, out=dw, accumulate=True)
for _ in range(6):
    torch.nn.functional.linear(x, w)
a, b = torch.randn(4096, 1024, device="cuda"), torch.randn(1024, 1024, device="cuda")
for _ in range(6):
    gemm.mm_nt(a, b)
h = torch.randn(128, 128, 32, 32, device="cuda")
for i in range(6):
    ops.dropout(h, 0.1, 1234 + i, 0)
torch.cuda.synchronize()
