"""Epoch driver + fused optimizer behind the `unlearn` plugin surface.

`iterative_unlearn` keeps the reference contract (Classification/unlearn/impl.py:54-127):
a per-epoch plugin ``f(data_loaders, model, criterion, optimizer, epoch, args, mask)``
is wrapped into ``method(data_loaders, model, criterion, args, mask=None)`` which owns the
optimizer (SGD: lr ``unlearn_lr``, ``momentum``, ``weight_decay``), the MultiStepLR
schedule (``decreasing_lr`` milestones, gamma 0.1) and the epoch loop.

What changes is *how a step executes*: the model's parameters are re-homed into one flat
arena and the reference's three per-tensor stages

    _apply_mask_to_grads -> optimizer.step() -> _restore_masked_params   (RL.py:134-140)

(~440 launches + 62 host syncs per step) become ONE
`salun_masked_sgd_step` launch inside `FusedMaskedSGD.step()`.
"""
from __future__ import annotations

import os
import time
from typing import Dict, Optional

import torch

from ... import ops
from ...dist import all_reduce_mean_, world_size
from ...flat import FlatArena, arena_of
from .. import utils


class FusedMaskedSGD(torch.optim.Optimizer):
    """SGD(momentum, weight_decay, dampening 0, no nesterov) over a FlatArena with the saliency
    mask folded in.  A torch.optim.Optimizer so LR schedulers / state_dict work; `param_groups[0]`
    carries lr / momentum / weight_decay like torch.optim.SGD's."""

    def __init__(self, arena: FlatArena, lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        self.arena = arena
        super().__init__(arena._params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.momentum_buffer = arena.new_like() if momentum != 0 else None
        self.mask_u8: Optional[torch.Tensor] = None
        self._first_step = True
        self.steps = 0

    def set_mask(self, mask_u8: Optional[torch.Tensor]) -> None:
        if mask_u8 is not None:
            assert mask_u8.dtype == torch.uint8 and mask_u8.numel() == self.arena.n
        self.mask_u8 = mask_u8

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: ARG002 - grads stay attached views
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        if world_size() > 1:  # data parallel: one all-reduce of the flat gradient (RCCL over xGMI)
            all_reduce_mean_(self.arena.grads)
        ops.masked_sgd_step(self.arena.params, self.arena.grads, self.momentum_buffer, self.mask_u8,
                            g["lr"], g["momentum"], g["weight_decay"], self._first_step)
        self._first_step = False
        self.steps += 1
        return loss


def plot_training_curve(training_result, save_dir, prefix):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # plotting is optional
        return
    for name, result in training_result.items():
        plt.plot(result, label=f"{name}_acc")
    plt.legend()
    plt.savefig(os.path.join(save_dir, prefix + "_train.png"))
    plt.close()


def save_unlearn_checkpoint(model, evaluation_result, args):
    """{save_dir}/{unlearn}checkpoint.pth.tar (+ ...eval_result.pth.tar), reference impl.py:21-30."""
    state = {"state_dict": model.state_dict(), "evaluation_result": evaluation_result}
    utils.save_checkpoint(state, False, args.save_dir, args.unlearn)
    utils.save_checkpoint(evaluation_result, False, args.save_dir, args.unlearn, filename="eval_result.pth.tar")


def load_unlearn_checkpoint(model, device, args):
    """-> (model, evaluation_result) or None.  (The reference also re-applies torch.nn.utils.prune
    masks found in the checkpoint, impl.py:38-40 — pruning is out of scope)"""
    ckpt = utils.load_checkpoint(device, args.save_dir, args.unlearn)
    if ckpt is None or ckpt.get("state_dict") is None:
        return None
    model.load_state_dict(ckpt["state_dict"])
    return model, ckpt.get("evaluation_result")


def _iterative_unlearn_impl(unlearn_iter_func):
    def _wrapped(data_loaders, model, criterion, args, mask: Optional[Dict[str, torch.Tensor]] = None, **kwargs):
        if getattr(args, "rewind_epoch", 0) != 0:
            raise NotImplementedError("weight rewinding belongs to the pruning baselines (out of scope)")
        milestones = [int(x) for x in str(args.decreasing_lr).split(",")]
        arena = arena_of(model)
        optimizer = FusedMaskedSGD(arena, args.unlearn_lr, momentum=args.momentum, weight_decay=args.weight_decay)
        if mask:
            optimizer.set_mask(arena.pack_mask(mask))
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=milestones, gamma=0.1)
        for epoch in range(0, args.unlearn_epochs):
            start_time = time.time()
            print("Epoch #{}, Learning rate: {}".format(epoch, optimizer.param_groups[0]["lr"]))
            unlearn_iter_func(data_loaders, model, criterion, optimizer, epoch, args, mask, **kwargs)
            scheduler.step()
            print("one epoch duration:{}".format(time.time() - start_time))

    _wrapped.__name__ = getattr(unlearn_iter_func, "__name__", "unlearn")
    _wrapped.__wrapped_iter__ = unlearn_iter_func
    return _wrapped


def iterative_unlearn(func):
    """usage:  @iterative_unlearn
               def func(data_loaders, model, criterion, optimizer, epoch, args, mask=None)"""
    return _iterative_unlearn_impl(func)
