"""The oracle standing in for the HIP kernels — TEST INFRASTRUCTURE ONLY.

The product path has no CPU fallback (`ops.py` rejects host tensors).  To execute the product's data-parallel HOST logic
on the CPU — `Diffusion.accumulate_saliency` / `unlearn_step`, `SD.train_scripts._saliency_mask` / `_unlearn`,
`FusedMaskedAdam` with its bucketed gradient reducer, the draws of draws.py — over world_size-2 `gloo`, a test worker
calls `install()`, which replaces the kernel entry points of `unlearn_saliency_amd.ops` IN THAT PROCESS by the CPU oracle
(oracle/) on torch host tensors.  Nothing under `unlearn_saliency_amd/` knows about this file.
"""
from __future__ import annotations

import numpy as np
import torch

import oracle


def _np(t: torch.Tensor) -> np.ndarray:
    assert t.device.type == "cpu" and t.is_contiguous()
    return t.detach().numpy()


def install() -> None:
    from unlearn_saliency_amd import ops

    def saliency_accumulate(acc, g, scale=1.0, sqnorm=None, max_norm=1.0):
        if sqnorm is not None:
            scale = oracle.clip_coef(float(sqnorm.item()), max_norm)
        oracle.saliency_accumulate(_np(acc), _np(g), scale)

    def grad_sqnorm(g, out=None):
        out = out if out is not None else torch.empty(1, dtype=torch.float32)
        out[0] = oracle.grad_sqnorm(_np(g))
        return out

    def masked_adam_step(p, g, m1, v, mask, lr, beta1, beta2, eps, weight_decay, step, sqnorm=None, max_norm=1.0,
                         gscale=1.0):
        ops.PARAM_EPOCH[0] += 1
        if sqnorm is not None:  # the kernel's rule: the clip coefficient replaces gscale
            gscale = oracle.clip_coef(float(sqnorm.item()), max_norm)
        oracle.masked_adam_step(_np(p), _np(g), _np(m1), _np(v), None if mask is None else _np(mask), gscale, lr, beta1,
                                beta2, eps, weight_decay, step)

    def qsample(x0, e, sqrt_ab, sqrt_1mab, t):
        return torch.from_numpy(oracle.qsample(_np(x0), _np(e), _np(sqrt_ab), _np(sqrt_1mab), _np(t)))

    def sqerr_loss(a, b, coef, want_per_sample=False, want_grad=True):
        loss, per, d = oracle.sqerr_loss(_np(a), _np(b), coef, want_grad)
        return (torch.tensor([loss], dtype=torch.float32), torch.from_numpy(per) if want_per_sample else None,
                torch.from_numpy(d) if want_grad else None)

    def mask_topk(acc, ks, out=None, flags=0, check=False):
        return [torch.from_numpy(m) for m in oracle.mask_topk(_np(acc), list(ks))]

    def mask_u8_to_i64(m):
        return m.to(torch.int64)

    def mask_i64_to_u8(m, out=None):
        r = (m != 0).to(torch.uint8)
        if out is not None:
            out.copy_(r)
            return out
        return r

    for name, fn in dict(saliency_accumulate=saliency_accumulate, grad_sqnorm=grad_sqnorm,
                         masked_adam_step=masked_adam_step, qsample=qsample, sqerr_loss=sqerr_loss, mask_topk=mask_topk,
                         mask_u8_to_i64=mask_u8_to_i64, mask_i64_to_u8=mask_i64_to_u8).items():
        setattr(ops, name, fn)
