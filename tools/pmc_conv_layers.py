"""Workload for the SQ counter passes over the fp32 convolution kernels (tools/pmc_multi.sh): each 3x3 stride-1 layer
shape of ResNet-18 (batch 256) a few times through the ring forward kernel, conv_igemm, the ring backward-weight kernel
and conv_wgrad_v."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops, ringpack

for N, C, H, K in ((256, 64, 32, 64), (256, 128, 16, 128), (256, 256, 8, 256), (256, 512, 4, 512)):
    x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05
    dy = torch.randn(N, K, H, H, device="cuda")
    imf = ops.conv3x3_pack(w, False)
    for _ in range(4):
        ops.conv3x3_packed(x, imf, K)
        ops.conv2d_forward(x, w, None, 1, 1, H, H)
        ops.conv2d_backward_weight(x, dy, w.shape, 1, 1)
        ops.conv2d_backward_weight(x, dy, w.shape, 1, 1, shared=True)
torch.cuda.synchronize()
