"""CIFAR-10 class-split loaders for the DDPM forget / remain sets (reference DDPM/datasets/__init__.py:120-177,
241-255).  The reference materialises both splits once as Python lists of (ToTensor image, label) with the
random flip frozen at materialisation time; here both splits are device-resident fp32 tensors built once
(flip drawn once, like the reference) and batches are index gathers on the GPU.  Falls back to the
counter-based synthetic CIFAR-shaped set when the dataset files are absent (no network on the GPU box)."""
from __future__ import annotations

import numpy as np
import torch

from ...Classification.dataset import _load_cifar10_files, have_cifar10, synthetic_cifar10
from ... import dist as sdist


class TensorLoader:
    """Shuffling batch iterator over device tensors (x: (N,3,32,32) fp32 in [0,1], c: (N,) int64)."""

    def __init__(self, x, c, batch_size, shuffle=True, rank=0, world_size=1):
        self.x, self.c, self.batch_size, self.shuffle = x, c, int(batch_size), shuffle
        self.rank, self.world_size = rank, world_size
        self.last_shard = None  # (lo, hi, b): this rank holds samples [lo, hi) of the global batch of b just yielded

    def __len__(self):
        return (len(self.x) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.x)
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        order = (torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)).to(self.x.device)
        for s in range(0, n, self.batch_size):
            idx = order[s:s + self.batch_size]
            b = idx.numel()
            lo, hi = sdist.balanced_slice(b, self.rank, self.world_size) if self.world_size > 1 else (0, b)
            self.last_shard = (lo, hi, b)
            idx = idx[lo:hi]
            yield self.x[idx], self.c[idx]


def get_forget_dataset(args, config, label_to_drop, device=None, synthetic=None):
    """-> (remain_loader, forget_loader) split by class `label_to_drop`."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    use_syn = synthetic if synthetic is not None else not have_cifar10(config.data.path)
    if use_syn:
        (x, y), _ = synthetic_cifar10()
    else:
        (x, y), _ = _load_cifar10_files(config.data.path)
    if getattr(config.data, "random_flip", True):  # frozen once, as in the reference's materialised lists
        flip = np.random.rand(len(x)) < 0.5
        x = np.where(flip[:, None, None, None], x[:, :, ::-1, :], x)
    xt = torch.from_numpy(np.ascontiguousarray(x)).to(device).permute(0, 3, 1, 2).float().div_(255).contiguous()
    yt = torch.from_numpy(y).to(device)
    forget = yt == int(label_to_drop)
    print(int((~forget).sum()), int(forget.sum()))
    rk, ws = sdist.rank(), sdist.world_size()
    bs = config.training.batch_size
    return (TensorLoader(xt[~forget], yt[~forget], bs, True, rk, ws),
            TensorLoader(xt[forget], yt[forget], bs, True, rk, ws))


def data_transform(config, X):
    """[0,1] -> model range (reference datasets/__init__.py:241-255)."""
    if config.data.uniform_dequantization:
        X = X / 256.0 * 255.0 + torch.rand_like(X) / 256.0
    if config.data.gaussian_dequantization:
        X = X + torch.randn_like(X) * 0.01
    if config.data.rescaled:
        X = 2 * X - 1.0
    elif config.data.logit_transform:
        lam = 1e-6
        X = lam + (1 - 2 * lam) * X
        X = torch.log(X) - torch.log1p(-X)
    if hasattr(config, "image_mean"):
        return X - config.image_mean.to(X.device)[None, ...]
    return X
