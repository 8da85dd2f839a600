"""Reverse-process samplers of the DDPM sub-project — same entry points and return values as the reference's
DDPM/functions/denoising.py:11-131 (`generalized_steps[_conditional]` = DDIM with eta, `ddpm_steps` /
`ddpm_step_conditional` = ancestral sampling with the posterior mean written through the clamped x0 estimate), used by
`Diffusion.sample_image` to evaluate an unlearned model.

One loop serves the four variants.  The alpha-bar table is built once on the device (the reference rebuilds
`cat -> cumprod -> index_select` and a fresh `ones(n) * i` timestep vector on the host for every step), per-step
coefficients are device scalars gathered from it, and the trajectory lists are optional: `keep="all"` reproduces the
reference's `(xs, x0_preds)` lists of host tensors, `keep="last"` holds only the current state on the device.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch


def alpha_bar_table(betas: torch.Tensor) -> torch.Tensor:
    """abar[t + 1] = prod_{s <= t} (1 - beta_s), abar[0] = 1 (the reference's `compute_alpha(beta, t)` reads entry t + 1,
    so a "next" timestep of -1 lands on 1)."""
    return torch.cat([betas.new_zeros(1), betas], dim=0).neg().add(1).cumprod(dim=0)


def compute_alpha(beta: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """abar_t as a (n,1,1,1) tensor — kept for callers of the reference helper (denoising.py:4-7)."""
    return alpha_bar_table(beta).index_select(0, t + 1).view(-1, 1, 1, 1)


def _loop(x: torch.Tensor, seq: Sequence[int], eps_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
          betas: torch.Tensor, ancestral: bool, eta: float, keep: str) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    seq = list(seq)
    prev = [-1] + seq[:-1]
    abar = alpha_bar_table(betas.to(x.device))
    n = x.size(0)
    xs, x0s = [x], []
    cur = x
    with torch.no_grad():
        for i, j in zip(reversed(seq), reversed(prev)):
            t = torch.full((n,), float(i), device=x.device)
            at, an = abar[i + 1], abar[j + 1]
            cur = cur.to(x.device)
            et = eps_fn(cur, t)
            if ancestral:
                # x0 from eps, clamped; posterior mean through x0; variance beta_t = 1 - abar_t / abar_{t-1}
                beta_t = 1 - at / an
                x0 = torch.clamp((1.0 / at).sqrt() * cur - (1.0 / at - 1).sqrt() * et, -1, 1)
                mean = ((an.sqrt() * beta_t) * x0 + ((1 - beta_t).sqrt() * (1 - an)) * cur) / (1.0 - at)
                noise = torch.randn_like(cur)
                nxt = mean + (0.0 if i == 0 else 1.0) * torch.exp(0.5 * beta_t.log()) * noise
            else:
                x0 = (cur - et * (1 - at).sqrt()) / at.sqrt()
                c1 = eta * ((1 - at / an) * (1 - an) / (1 - at)).sqrt()
                c2 = ((1 - an) - c1 ** 2).sqrt()
                nxt = an.sqrt() * x0 + c1 * torch.randn_like(cur) + c2 * et
            if keep == "all":
                x0s.append(x0.to("cpu"))
                xs.append(nxt.to("cpu"))
            else:
                x0s, xs = [x0], [nxt]
            cur = nxt
    return xs, x0s


def generalized_steps(x, seq, model, b, keep: str = "all", **kwargs):
    return _loop(x, seq, lambda xt, t: model(xt, t), b, False, kwargs.get("eta", 0), keep)


def ddpm_steps(x, seq, model, b, keep: str = "all", **kwargs):
    return _loop(x, seq, lambda xt, t: model(xt, t.float()), b, True, 0.0, keep)


def generalized_steps_conditional(x, c, seq, model, b, cond_scale=3.0, keep: str = "all", **kwargs):
    return _loop(x, seq, lambda xt, t: model(xt, t, c, cond_scale=cond_scale, mode="test"), b, False,
                 kwargs.get("eta", 0), keep)


def ddpm_step_conditional(x, c, seq, model, b, cond_scale, keep: str = "all", **kwargs):
    return _loop(x, seq, lambda xt, t: model(xt, t.float(), c, cond_scale=cond_scale, mode="test"), b, True, 0.0, keep)
