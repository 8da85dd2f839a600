"""Fused optimizers over a FlatArena: one HIP launch per step for the whole model.

Both are `torch.optim.Optimizer`s (LR schedulers, `param_groups`, `state_dict` keep working) whose
`step()` is: [multi-GPU: one all-reduce(mean) of the flat gradient] -> one kernel that applies
the saliency mask, the update rule and (SGD) the reference's restore of masked-out weights.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .dist import all_reduce_mean_, collectives_on
from .flat import FlatArena


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, arena: FlatArena, defaults: dict):
        self.arena = arena
        super().__init__(arena._params, defaults)
        self.mask_u8: Optional[torch.Tensor] = None
        self.steps = 0
        self.overlap_buckets = 4
        self._reducer = None
        if collectives_on():
            from .dist import BucketedGradReducer
            self._reducer = BucketedGradReducer(arena, self.overlap_buckets)

    def close(self) -> None:
        """Detach from the arena: removes the gradient-bucket hooks this optimizer registered on the parameters (a
        second optimizer on the same model would otherwise leave the old reducer firing extra all-reduces on every
        backward).  Called by the epoch driver when the optimizer is done; idempotent."""
        if self._reducer is not None:
            self._reducer.remove()
            self._reducer = None
        self._closed = True

    def __del__(self):
        try:
            if getattr(self, "_reducer", None) is not None:
                self._reducer.remove()
        except Exception:
            pass

    # ---- torch.optim-layout state (checkpoints in the reference's [model, optimizer, step] format resume)
    _STATE_VECTORS: tuple = ()

    def state_dict(self):
        """Same layout as the torch.optim optimizer this one replaces: `state[i]` holds per-parameter tensors (views
        of the flat state vectors, cloned) so a checkpoint written here loads into torch.optim.SGD / Adam and back."""
        sd = super().state_dict()
        state = {}
        started = self.steps > 0
        if started:
            for i, (o, k, shp) in enumerate(zip(self.arena.offsets, self.arena.numels, self.arena.shapes)):
                ent = {}
                for key, attr in self._STATE_VECTORS:
                    vec = getattr(self, attr, None)
                    if vec is not None:
                        ent[key] = vec[o:o + k].view(shp).clone()
                if self._STEP_KEY:
                    ent["step"] = torch.tensor(float(self.steps))
                state[i] = ent
        sd["state"] = state
        sd["salun_steps"] = self.steps
        return sd

    _STEP_KEY = False

    def load_state_dict(self, sd):
        state = sd.get("state", {})
        groups = sd.get("param_groups")
        if groups:
            for g, src in zip(self.param_groups, groups):
                for key, val in src.items():
                    if key != "params":
                        g[key] = val
        steps = int(sd.get("salun_steps", 0))
        for i, (o, k, shp) in enumerate(zip(self.arena.offsets, self.arena.numels, self.arena.shapes)):
            ent = state.get(i, state.get(str(i)))
            if not ent:
                continue
            for key, attr in self._STATE_VECTORS:
                vec = getattr(self, attr, None)
                if vec is not None and key in ent:
                    vec[o:o + k].copy_(ent[key].reshape(-1).to(vec.device, vec.dtype))
            if "step" in ent:
                steps = max(steps, int(float(ent["step"])))
        self.steps = steps
        self._after_load()

    def _after_load(self) -> None:
        pass

    def set_mask(self, mask_u8: Optional[torch.Tensor]) -> None:
        """Flat u8 0/1 vector (FlatArena.pack_mask) or None for an unmasked update."""
        if mask_u8 is not None:
            assert mask_u8.dtype == torch.uint8 and mask_u8.numel() == self.arena.n
        self.mask_u8 = mask_u8

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: ARG002 - grads stay attached views
        self.arena.zero_grad()

    def _sync_grads(self) -> None:
        """Data parallel: mean of the flat gradient over ranks.  By default the all-reduce is cut into a few large
        slices that start during backward (dist.BucketedGradReducer); `overlap_buckets = 0` falls back to one
        collective over the whole vector after backward."""
        if not collectives_on():
            return
        if self.overlap_buckets > 0:
            if self._reducer is None:  # first step: hooks were not installed before this backward ran
                from .dist import BucketedGradReducer
                self._reducer = BucketedGradReducer(self.arena, self.overlap_buckets)
            self._reducer.finish()
        else:
            all_reduce_mean_(self.arena.grads)


class FusedMaskedSGD(_FlatOptimizer):
    """torch.optim.SGD(momentum, weight_decay, dampening=0, nesterov=False) semantics
    (Classification/unlearn/impl.py:68-73) + `_apply_mask_to_grads` + `_restore_masked_params`
    (Classification/unlearn/RL.py:11-34) in one `salun_masked_sgd_step` launch."""

    _STATE_VECTORS = (("momentum_buffer", "momentum_buffer"),)

    def __init__(self, arena: FlatArena, lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        super().__init__(arena, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.momentum_buffer = arena.new_like() if momentum != 0 else None
        self._first_step = True

    def _after_load(self) -> None:
        self._first_step = self.steps == 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._sync_grads()
        ops.masked_sgd_step(self.arena.params, self.arena.grads, self.momentum_buffer, self.mask_u8,
                            g["lr"], g["momentum"], g["weight_decay"], self._first_step)
        self._first_step = False
        self.steps += 1
        return loss


class FusedMaskedAdam(_FlatOptimizer):
    """clip_grad_norm_(max_norm) -> mask multiply -> torch.optim.Adam(amsgrad=False) step
    (DDPM/runners/diffusion.py:582-593, DDPM/functions/__init__.py:9-18) as two launches:
    `salun_grad_sqnorm` (deterministic reduction, result stays on the device) and
    `salun_masked_adam_step` (reads the norm from device memory: no host sync anywhere)."""

    _STATE_VECTORS = (("exp_avg", "exp_avg"), ("exp_avg_sq", "exp_avg_sq"))
    _STEP_KEY = True

    def __init__(self, arena: FlatArena, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, amsgrad: bool = False, grad_clip: Optional[float] = None):
        if amsgrad:
            raise NotImplementedError("amsgrad is off in every reference config (cifar10_saliency_unlearn.yml:54)")
        super().__init__(arena, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.exp_avg = arena.new_like()
        self.exp_avg_sq = arena.new_like()
        self.grad_clip = grad_clip
        self._sqnorm = torch.zeros(1, dtype=torch.float32, device=arena.device)
        self._step_dev = None  # device-resident step counter (use_device_step)
        self._coef = None

    def use_device_step(self) -> None:
        """Keep Adam's step count (hence its bias corrections) on the device (salun_adam_coefficients advances it and
        derives the two step-dependent scalars there): the kernel arguments of a step are then the same every step.
        `steps` on the host keeps counting `step()` calls."""
        if self._step_dev is None:
            self._step_dev = torch.tensor([self.steps], dtype=torch.int64, device=self.arena.device)
            self._coef = torch.zeros(2, dtype=torch.float32, device=self.arena.device)

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Arms clipping for the next step() and returns the squared norm (device tensor, no sync).
        Mirrors the call order of the reference loop: clip, then mask, then step."""
        self.grad_clip = max_norm
        return self._sqnorm

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._sync_grads()
        self.steps += 1
        sq = None
        if self.grad_clip is not None:
            sq = ops.grad_sqnorm(self.arena.grads, self._sqnorm)
        if self._step_dev is not None:
            ops.adam_coefficients(self._step_dev, g["lr"], g["betas"][0], g["betas"][1], self._coef)
            ops.masked_adam_step_coef(self.arena.params, self.arena.grads, self.exp_avg, self.exp_avg_sq, self.mask_u8,
                                      self._coef, g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], sqnorm=sq,
                                      max_norm=self.grad_clip if self.grad_clip is not None else 1.0)
            return loss
        ops.masked_adam_step(self.arena.params, self.arena.grads, self.exp_avg, self.exp_avg_sq, self.mask_u8,
                             g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.steps,
                             sqnorm=sq, max_norm=self.grad_clip if self.grad_clip is not None else 1.0)
        return loss
