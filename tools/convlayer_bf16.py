"""One SD layer on the bf16 convolution kernels, a few launches each (PMC / trace target).
python tools/convlayer_bf16.py H C K R stride [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops
H, C, K, R, st = (int(v) for v in sys.argv[1:6])
it = int(sys.argv[6]) if len(sys.argv) > 6 else 5
x = torch.randn(8, H, H, C, device="cuda").to(torch.bfloat16)
w = torch.randn(K, C, R, R, device="cuda") / (C * R * R) ** 0.5
wp = ops.conv2d_bf16_pack(w)
y = ops.conv2d_bf16_forward(x, wp, R, st, R // 2)
dy = torch.randn_like(y)
for _ in range(it):
    ops.conv2d_bf16_forward(x, wp, R, st, R // 2)
    ops.conv2d_bf16_backward_data(dy, wp, tuple(x.shape), R, st, R // 2)
    ops.conv2d_bf16_backward_weight(x, dy, tuple(w.shape), st, R // 2)
torch.cuda.synchronize()
