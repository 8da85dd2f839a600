"""ε-prediction losses (reference DDPM/functions/losses.py:21-46).

`noise_estimation_loss_conditional` keeps the reference's signature and registry key; the q-sample
(x_t = x0·√ā_t + e·√(1−ā_t)) and the squared-error reduction (+ its gradient, produced in the same pass)
run as the HIP kernels `salun_qsample` / `salun_sqerr_loss` instead of ~8 small ATen launches.
"""
from __future__ import annotations

import torch

from ... import ops

_TABLE_CACHE: dict = {}


def alpha_tables(b: torch.Tensor):
    """(√ā, √(1−ā)) for a β schedule tensor; `(1 - b).cumprod(0)` exactly as the reference computes it per
    call (losses.py:31), cached per schedule tensor."""
    key = (b.data_ptr(), b.numel(), str(b.device))
    hit = _TABLE_CACHE.get(key)
    if hit is None:
        a = (1 - b).cumprod(dim=0)
        hit = (a.sqrt().contiguous(), (1.0 - a).sqrt().contiguous())
        _TABLE_CACHE.clear()
        _TABLE_CACHE[key] = hit
    return hit


def q_sample(x0: torch.Tensor, t: torch.Tensor, e: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    sa, sb = alpha_tables(b)
    return ops.qsample(x0.contiguous(), e.contiguous(), sa, sb, t.to(torch.int64).contiguous())


def noise_estimation_loss_conditional(model, x0: torch.Tensor, t: torch.LongTensor, c: torch.LongTensor,
                                      e: torch.Tensor, b: torch.Tensor, cond_drop_prob=0.1, keepdim=False):
    x = q_sample(x0, t, e, b)
    output = model(x, t.float(), c, cond_drop_prob=cond_drop_prob, mode="train")
    if keepdim:  # per-sample losses (used by the Fisher computation): autograd through plain ops
        return (e - output).square().sum(dim=(1, 2, 3))
    return ops.eps_mse(e, output)


loss_registry_conditional = {"simple": noise_estimation_loss_conditional}
