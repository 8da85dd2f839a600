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


# ------------------------------------------------------------------ SD pieces (plain torch, any device / dtype)
class PlainLDM:
    """The slice of the reference's LatentDiffusion the SalUn scripts call, in plain torch ops around a given U-Net:
    q_sample (SD/ldm/models/diffusion/ddpm.py:424-430), apply_model (:1121), shared_step -> p_losses with
    logvar = 0, l_simple_weight = 1, original_elbo_weight = 0 (:1093-1109, :1286-1319).  Schedule: the LDM "linear"
    schedule linspace(sqrt(start), sqrt(end))**2 in float64 (ldm/modules/diffusionmodules/util.py:24-30)."""

    def __init__(self, unet, timesteps=1000, linear_start=0.00085, linear_end=0.0120):
        import numpy as np
        self.unet, self.num_timesteps = unet, int(timesteps)
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
        ac = np.cumprod(1.0 - betas, axis=0)
        p = next(unet.parameters())
        self.device = p.device
        self.sqrt_ac = torch.tensor(np.sqrt(ac), dtype=torch.float32, device=p.device)
        self.sqrt_1mac = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32, device=p.device)

    def q_sample(self, x_start, t, noise):
        a = self.sqrt_ac.index_select(0, t).view(-1, 1, 1, 1)
        b = self.sqrt_1mac.index_select(0, t).view(-1, 1, 1, 1)
        return a * x_start + b * noise

    def apply_model(self, x_noisy, t, cond):
        return self.unet(x_noisy, t, context=cond)

    def shared_step(self, z, c):
        t = torch.randint(0, self.num_timesteps, (z.shape[0],), device=self.device).long()
        noise = torch.randn_like(z)
        return nn.MSELoss()(self.apply_model(self.q_sample(z, t, noise), t, c), noise)


def sd_saliency_gradients(ldm: PlainLDM, batches, c_guidance) -> Dict[str, torch.Tensor]:
    """SD/train-scripts/generate_mask.py:130-174 (generate_nsfw_mask; :24-69 for generate_mask): per batch
    loss = -MSE(noise, (1+g) eps(z_t, c) - g eps(z_t, null)), uniform t, per-tensor `gradients[name] += grad`.
    (The reference draws an unused `t` first, :141-143; not reproduced: it only advances the generator.)"""
    unet = ldm.unet
    unet.eval()
    gradients = {name: 0 for name, _ in unet.named_parameters()}
    for z, c_forget, c_null in batches:
        unet.zero_grad()
        t = torch.randint(0, ldm.num_timesteps, (z.shape[0],), device=ldm.device).long()
        noise = torch.randn_like(z)
        z_noisy = ldm.q_sample(z, t, noise)
        preds = (1 + c_guidance) * ldm.apply_model(z_noisy, t, c_forget) - c_guidance * ldm.apply_model(z_noisy, t, c_null)
        loss = -nn.MSELoss()(noise, preds)
        loss.backward()
        with torch.no_grad():
            for name, param in unet.named_parameters():
                if param.grad is not None:
                    gradients[name] = gradients[name] + param.grad.detach().clone()
    return gradients


def sd_unlearn(ldm: PlainLDM, forget_batches, remain_batches, alpha, lr, mask, train_method="full", epochs=1):
    """SD/train-scripts/nsfw_removal.py:60-150 (random_label.py:58-139 is the same body): remain loss via shared_step,
    forget/pseudo outputs on one noisy latent, loss = MSE(forget_out, pseudo_out.detach()) + alpha * remain_loss,
    per-tensor `p.grad *= mask[name]`, torch.optim.Adam(lr).  Returns (losses, optimizer)."""
    unet = ldm.unet
    params = [p for n, p in unet.named_parameters() if train_method == "full" or "attn2" in n]
    unet.train()
    opt = torch.optim.Adam(params, lr=lr)
    losses = []
    for _ in range(epochs):
        remain_iter = iter(remain_batches)
        for z_f, c_forget, c_pseudo in forget_batches:
            opt.zero_grad()
            unet.zero_grad()
            try:
                z_r, c_r = next(remain_iter)
            except StopIteration:
                remain_iter = iter(remain_batches)
                z_r, c_r = next(remain_iter)
            remain_loss = ldm.shared_step(z_r, c_r)
            t = torch.randint(0, ldm.num_timesteps, (z_f.shape[0],), device=ldm.device).long()
            noise = torch.randn_like(z_f)
            z_noisy = ldm.q_sample(z_f, t, noise)
            forget_out = ldm.apply_model(z_noisy, t, c_forget)
            pseudo_out = ldm.apply_model(z_noisy, t, c_pseudo).detach()
            loss = nn.MSELoss()(forget_out, pseudo_out) + alpha * remain_loss
            loss.backward()
            losses.append(float(loss.item()))
            if mask:
                for n, p in unet.named_parameters():
                    if p.grad is not None:
                        p.grad *= mask[n].to(p.grad.device)
            opt.step()
    unet.eval()
    return losses, opt


def sd_proximal_unlearn(ldm: PlainLDM, forget_batches, remain_batches, alpha, lr, mask_ratio, n_frozen=0, epochs=1,
                        train_method="full"):
    """SD/train-scripts/proximal_gradient.py:76-186: the `sd_unlearn` body without a saliency mask, followed after every
    optimizer step by the proximal pull towards the initial weights.  The reference ranks |theta - theta_0| over
    `model.parameters()` of the WHOLE LatentDiffusion (:66-72): `n_frozen` zeros stand for its frozen first stage and
    text encoder (they never move), appended to the U-Net's differences before the top-k (:157-167); the three-way update
    (:169-182) is `param -= init; param[larger] -= thr; param[smaller] += thr; param[between] = 0; param += init`."""
    unet = ldm.unet
    params = [p for n, p in unet.named_parameters() if train_method == "full" or "attn2" in n]
    unet.train()
    opt = torch.optim.Adam(params, lr=lr)
    init = torch.cat([p.detach().reshape(-1) for p in unet.parameters()]).clone()
    n_params = init.numel() + int(n_frozen)
    per_epoch = len(forget_batches) + len(remain_batches)
    total_steps = epochs * per_epoch
    losses = []
    for epoch in range(epochs):
        remain_iter = iter(remain_batches)
        for i, (z_f, c_forget, c_pseudo) in enumerate(forget_batches):
            opt.zero_grad()
            try:
                z_r, c_r = next(remain_iter)
            except StopIteration:
                remain_iter = iter(remain_batches)
                z_r, c_r = next(remain_iter)
            remain_loss = ldm.shared_step(z_r, c_r)
            t = torch.randint(0, ldm.num_timesteps, (z_f.shape[0],), device=ldm.device).long()
            noise = torch.randn_like(z_f)
            z_noisy = ldm.q_sample(z_f, t, noise)
            loss = nn.MSELoss()(ldm.apply_model(z_noisy, t, c_forget), ldm.apply_model(z_noisy, t, c_pseudo).detach()) \
                + alpha * remain_loss
            loss.backward()
            losses.append(float(loss.item()))
            opt.step()
            with torch.no_grad():
                ratio = int(mask_ratio * ((total_steps - (epoch * per_epoch + i + 1)) / total_steps * n_params))
                cur = torch.cat([p.reshape(-1) for p in unet.parameters()] + [torch.zeros(int(n_frozen))])
                cur[:init.numel()] -= init
                cur.abs_().neg_()
                threshold = -torch.topk(cur, ratio)[0][-1]
                cnt = 0
                for p in unet.parameters():
                    ip = init[cnt:cnt + p.numel()].view(p.shape)
                    p -= ip
                    larger, smaller = p > threshold, p < -threshold
                    between = ~(larger | smaller)
                    p[larger] -= threshold
                    p[smaller] += threshold
                    p[between] = 0
                    p += ip
                    cnt += p.numel()
    unet.eval()
    return losses, opt
