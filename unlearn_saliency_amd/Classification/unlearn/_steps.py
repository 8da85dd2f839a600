"""One optimisation pass over a loader — the loop body every SGD-based unlearning plugin
shares (reference RL.py:123-176, GA.py:113-153, FT.py:115-168 repeat it with a different
label rule / loss sign / regulariser).  The step itself is three calls:

    optimizer.zero_grad()   one memset of the flat gradient
    loss.backward()         autograd accumulates into the flat views
    optimizer.step()        one fused mask * SGD-momentum * restore launch (+ DP all-reduce)

Meters are accumulated on the device and read back only when something is printed, so the
pass issues no per-step host synchronisation (the reference syncs ~64 times per step).
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch

import contextlib

from .. import utils
from ... import dist as sdist
from ... import resblock


def l1_regularization(model) -> torch.Tensor:
    """||theta||_1 over all parameters (reference trainer/train.py:10-14)."""
    return torch.norm(torch.cat([p.view(-1) for p in model.parameters()]), p=1)


def run_pass(loader, model, criterion, optimizer, epoch: int, args, *,
             label_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
             loss_sign: float = 1.0, l1_alpha: float = 0.0, track: bool = True,
             losses: Optional[utils.AverageMeter] = None, top1: Optional[utils.AverageMeter] = None,
             loader_len: Optional[int] = None, warmup_steps_per_epoch: Optional[int] = None,
             batch_label_fn: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
             after_step: Optional[Callable[[int], None]] = None, step_offset: int = 0):
    """Run every batch of `loader` through forward/backward/fused-step.

    label_fn   maps the batch's true targets (CPU or device) to the targets used in the loss
               (RL: fresh uniform random labels); None keeps them.
    loss_sign  -1 for gradient ascent.      l1_alpha  weight of the l1 penalty (FT_l1 / GA_l1).
    track      update `losses` / `top1` (sample-weighted) like the reference's retain loop.
    batch_label_fn  (device image, device true target) -> labels for the loss; accuracy is still measured against
               the TRUE targets (boundary_sh.py:99-116: loss on the adversarial neighbour label, prec1 on `target`).
    after_step      called with the loop index after every optimizer step (RL_proximal's soft-threshold).
    """
    dev = next(model.parameters()).device
    loader_len = len(loader) if loader_len is None else loader_len
    loss_sum = torch.zeros((), device=dev, dtype=torch.float64)
    hit_sum = torch.zeros((), device=dev, dtype=torch.float64)
    seen = 0
    start = time.time()
    for i, (image, target) in enumerate(loader):
        if warmup_steps_per_epoch is not None and epoch < args.warmup:
            utils.warmup_lr(epoch, i + 1, optimizer, one_epoch_step=warmup_steps_per_epoch, args=args)
        shard = getattr(loader, "last_shard", None) if sdist.world_size() > 1 else None
        if label_fn is not None:
            if shard is not None:
                # data parallel: the labels are drawn once per GLOBAL batch on every rank (same generator state),
                # then sliced — the reference draws them for the whole batch (RL.py:125)
                lo, hi, b = shard
                target = label_fn(target.new_empty((b,) + tuple(target.shape[1:])))[lo:hi]
            else:
                target = label_fn(target)
        image = image.to(dev, non_blocking=True)
        target = target.to(dev, non_blocking=True)

        if image.size(0) == 0:  # ragged tail smaller than the world size: contribute a zero gradient
            optimizer.zero_grad()
            optimizer.step()
            if after_step is not None:
                after_step(i + step_offset)
            continue
        loss_target = target if batch_label_fn is None else batch_label_fn(image, target)
        output = model(image)
        loss = criterion(output, loss_target)
        if loss_sign != 1.0:
            loss = loss_sign * loss
        if shard is not None:  # count-weight the shard mean so that AVG over ranks is the global batch mean
            loss = loss * sdist.shard_loss_scale(loader)
        if l1_alpha:
            loss = loss + l1_alpha * l1_regularization(model)

        optimizer.zero_grad()
        # the l1 term gives every parameter a second gradient producer (AccumulateGrad on the main stream): the
        # convolution kernels must then accumulate on the main stream too (resblock.overlap_disabled)
        with (resblock.overlap_disabled() if l1_alpha else contextlib.nullcontext()):
            loss.backward()
        optimizer.step()
        if after_step is not None:
            after_step(i + step_offset)

        if track:
            n = image.size(0)
            with torch.no_grad():
                loss_sum += loss.detach().double() * n
                hit_sum += (output.detach().argmax(dim=1) == target).sum().double()
            seen += n
            if (i + 1) % args.print_freq == 0:
                end = time.time()
                print("Epoch: [{0}][{1}/{2}]\tLoss ({3:.4f})\tAccuracy ({4:.3f})\tTime {5:.2f}".format(
                    epoch, i, loader_len, float(loss_sum.item()) / seen, float(hit_sum.item()) * 100.0 / seen,
                    end - start))
                start = time.time()
    if track and seen:
        if losses is not None:
            losses.update(float(loss_sum.item()) / seen, seen)
        if top1 is not None:
            top1.update(float(hit_sum.item()) * 100.0 / seen, seen)
