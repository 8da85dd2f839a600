"""Un-fused, per-tensor PyTorch-CPU restatement of the reference's hot loops — TEST INFRASTRUCTURE ONLY.

This is the op sequence the reference executes (stock ATen ops launched per parameter tensor from a
Python loop), restated so it can be (i) checked against the golden vectors captured from the imported
reference and (ii) timed on the GPU box's host cores as ``cpu_baseline`` (kind "port") — the reference's
own Python files never travel to the GPU box.  Each function cites the reference lines it follows.
Used only by tests/ and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch
import torch.nn as nn


# ----------------------------------------------------------------- Phase A (mask generation)
def save_gradient_ratio_cpu(forget_batches, model: nn.Module, criterion, ratios) -> Dict[float, Dict[str, torch.Tensor]]:
    """Classification/generate_mask.py:14-80 without the torch.save: per-tensor `gradients[name] += grad`,
    abs_, then per ratio: -cat -> argsort -> argsort -> per-tensor (ranks < k) int64 masks."""
    gradients = {name: 0 for name, _ in model.named_parameters()}
    model.eval()
    for image, target in forget_batches:
        loss = -criterion(model(image), target)
        model.zero_grad()
        loss.backward()
        with torch.no_grad():
            for name, param in model.named_parameters():
                if param.grad is not None:
                    gradients[name] += param.grad.data
    return masks_from_gradients_cpu(gradients, ratios)


def masks_from_gradients_cpu(gradients: Dict[str, torch.Tensor], ratios) -> Dict[float, Dict[str, torch.Tensor]]:
    """generate_mask.py:46-80 on a ready gradient dict (stable argsort so ties are defined)."""
    with torch.no_grad():
        for name in gradients:
            gradients[name] = torch.abs_(gradients[name])
    out = {}
    for r in ratios:
        all_elements = -torch.cat([t.flatten() for t in gradients.values()])
        k = int(len(all_elements) * r)
        positions = torch.argsort(all_elements, stable=True)
        ranks = torch.argsort(positions, stable=True)
        hard, start = {}, 0
        for key, t in gradients.items():
            n = t.numel()
            tr = ranks[start:start + n]
            m = torch.zeros_like(tr)
            m[tr < k] = 1
            hard[key] = m.reshape(t.shape)
            start += n
        out[r] = hard
    return out


# ------------------------------------------------------------------ Phase B (masked RL step)
def apply_mask_to_grads(model, mask):
    """Classification/unlearn/RL.py:11-14."""
    for name, param in model.named_parameters():
        if param.grad is not None:
            param.grad *= mask[name]


def restore_masked_params(model, mask, theta0, optimizer):
    """Classification/unlearn/RL.py:17-34."""
    with torch.no_grad():
        for name, param in model.named_parameters():
            if name not in mask:
                continue
            m = mask[name].to(device=param.device, dtype=param.dtype)
            inv = 1 - m
            if torch.count_nonzero(inv) == 0:
                continue
            param.data.mul_(m).add_(theta0[name] * inv)
            st = optimizer.state.get(param, None)
            if st is not None and "momentum_buffer" in st:
                st["momentum_buffer"].mul_(m)


def rl_step_cpu(model, criterion, optimizer, image, target, mask, theta0, timers: Optional[dict] = None):
    """One unlearning step as the reference runs it (RL.py:128-140): forward, loss, zero_grad, backward,
    mask-multiply, SGD.step, restore.  `timers` (optional) accumulates the per-stage wall time."""
    t0 = time.perf_counter()
    loss = criterion(model(image), target)
    optimizer.zero_grad()
    loss.backward()
    t1 = time.perf_counter()
    if mask:
        apply_mask_to_grads(model, mask)
    t2 = time.perf_counter()
    optimizer.step()
    t3 = time.perf_counter()
    if mask:
        restore_masked_params(model, mask, theta0, optimizer)
    t4 = time.perf_counter()
    if timers is not None:
        for k, v in (("fwd_bwd", t1 - t0), ("mask_mul", t2 - t1), ("sgd_step", t3 - t2), ("restore", t4 - t3)):
            timers[k] = timers.get(k, 0.0) + v
    return loss


# ------------------------------------------------------------------ DDPM pieces
def eps_mse_cpu(e, out):
    """DDPM/functions/losses.py:37."""
    return (e - out).square().sum(dim=(1, 2, 3)).mean(dim=0)


def qsample_cpu(x0, e, betas, t):
    """DDPM/functions/losses.py:31-32."""
    a = (1 - betas).cumprod(dim=0).index_select(0, t).view(-1, 1, 1, 1)
    return x0 * a.sqrt() + e * (1.0 - a).sqrt()


def masked_adam_step_cpu(model, optimizer, mask, grad_clip=1.0):
    """DDPM/runners/diffusion.py:582-593: clip_grad_norm_ -> per-tensor mask multiply -> Adam.step."""
    torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
    if mask:
        for name, param in model.named_parameters():
            if param.grad is not None:
                param.grad *= mask[name].to(param.grad.device)
    optimizer.step()
