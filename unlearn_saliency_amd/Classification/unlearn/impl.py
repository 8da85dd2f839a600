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
`salun_masked_sgd_step` launch inside `FusedMaskedSGD.step()` (unlearn_saliency_amd/optim.py).
"""
from __future__ import annotations

import os
import time
from typing import Dict, Optional

import torch

from ... import hostperf
from ...flat import arena_of
from ...optim import FusedMaskedSGD
from .. import utils


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
        if mask and not getattr(unlearn_iter_func, "_ignores_mask", False):
            optimizer.set_mask(arena.pack_mask(mask))
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=milestones, gamma=0.1)
        try:
            hostperf.freeze_gc()
            for epoch in range(0, args.unlearn_epochs):
                start_time = time.time()
                print("Epoch #{}, Learning rate: {}".format(epoch, optimizer.param_groups[0]["lr"]))
                unlearn_iter_func(data_loaders, model, criterion, optimizer, epoch, args, mask, **kwargs)
                scheduler.step()
                print("one epoch duration:{}".format(time.time() - start_time))
        finally:
            optimizer.close()  # drop the data-parallel gradient hooks this optimizer put on the parameters

    _wrapped.__name__ = getattr(unlearn_iter_func, "__name__", "unlearn")
    _wrapped.__wrapped_iter__ = unlearn_iter_func
    return _wrapped


def iterative_unlearn(func):
    """usage:  @iterative_unlearn
               def func(data_loaders, model, criterion, optimizer, epoch, args, mask=None)"""
    return _iterative_unlearn_impl(func)
