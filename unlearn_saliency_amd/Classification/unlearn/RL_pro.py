"""RL with a proximal (soft-threshold) step (reference Classification/unlearn/RL_pro.py:8-157).

Each epoch: the forget set is relabelled with uniform random labels (`np.random.randint`), merged with the
retain set and walked in one shuffled pass; after EVERY SGD step the weights are pulled back towards the
epoch's starting point theta0 by a soft threshold whose level is the `ratio`-th smallest |theta - theta0|,

    ratio = int(mask_ratio * ((total_steps - (epoch * steps_per_epoch + 1)) / total_steps * n_params))

— i.e. at least `ratio` weights are reset to theta0 exactly and the rest shrink by the threshold.  The reference
does this with a concat of 62 tensors, a full `topk`, two `where`s and a per-tensor copy back per step; here it is
`ops.proximal_step` on the flat arena: diff -> radix select (the top-k machinery of the saliency mask) ->
soft-threshold, threshold never leaving the device.  `mask` is ignored, as in the reference.
The reference reads `args.mask_ratio`, which its arg_parser never defines; here it is the
`--mask_ratio` flag.
"""
import numpy as np
import torch

from ... import ops
from .. import utils
from ..dataset import ArrayDataset, BatchLoader
from ._steps import run_pass
from .impl import iterative_unlearn


def _merged_loader(forget_loader, retain_loader, forget_targets, batch_size):
    fd, rd = forget_loader.dataset, retain_loader.dataset
    if not (isinstance(fd, ArrayDataset) and isinstance(rd, ArrayDataset)):
        raise TypeError("RL_proximal merges the forget and retain sets: both loaders must wrap an ArrayDataset")
    merged = ArrayDataset(np.concatenate([fd.data, rd.data]),
                          np.concatenate([np.asarray(forget_targets), np.asarray(rd.targets)]),
                          transform=fd.transform)
    kw = {}
    if isinstance(forget_loader, BatchLoader):
        kw = dict(device_resident=forget_loader.device_resident, device=forget_loader.device,
                  rank=forget_loader.rank, world_size=forget_loader.world_size)
    return BatchLoader(merged, batch_size, True, **kw)


def _rl_proximal(data_loaders, model, criterion, optimizer, epoch, args, mask=None):
    forget_loader, retain_loader = data_loaders["forget"], data_loaders["retain"]
    mask_ratio = getattr(args, "mask_ratio", None)
    if mask_ratio is None:
        raise AttributeError("'Namespace' object has no attribute 'mask_ratio'  [pass --mask_ratio]")
    arena = optimizer.arena
    init_params = arena.params.clone()  # theta0 of THIS epoch (RL_pro.py:16)
    n_params = arena.n
    steps_per_epoch = len(forget_loader) + len(retain_loader)
    total_steps = args.unlearn_epochs * steps_per_epoch
    scratch = torch.empty_like(init_params)
    scratch_mask = torch.empty(n_params, dtype=torch.uint8, device=init_params.device)
    if epoch < args.warmup:
        raise NameError("name 'i' is not defined  [reference RL_pro.py:35-37: warmup>0 is unusable; keep --warmup 0]")
    losses, top1 = utils.AverageMeter(), utils.AverageMeter()
    model.train()

    def prox(ratio_step):
        def _after(i):
            ratio = int(mask_ratio * ((total_steps - ratio_step(i)) / total_steps * n_params))
            ops.proximal_step(arena.params, init_params, ratio, scratch, scratch_mask)
        return _after

    if args.dataset in ("cifar10", "cifar100", "TinyImagenet"):
        targets = np.random.randint(0, args.num_classes, np.asarray(forget_loader.dataset.targets).shape)
        train_loader = _merged_loader(forget_loader, retain_loader, targets, args.batch_size)
        # RL_pro.py:51: the ratio does not advance inside the epoch in this branch
        run_pass(train_loader, model, criterion, optimizer, epoch, args, track=True, losses=losses, top1=top1,
                 loader_len=steps_per_epoch, after_step=prox(lambda i: epoch * steps_per_epoch + 1),
                 step_offset=len(forget_loader))
    elif args.dataset == "svhn":
        run_pass(forget_loader, model, criterion, optimizer, epoch, args, track=False, loader_len=steps_per_epoch,
                 label_fn=lambda t: torch.randint(0, args.num_classes, t.shape),
                 after_step=prox(lambda i: epoch * steps_per_epoch + 1))
        run_pass(retain_loader, model, criterion, optimizer, epoch, args, track=True, losses=losses, top1=top1,
                 loader_len=steps_per_epoch, after_step=prox(lambda i: epoch * steps_per_epoch + i + 1))
    else:
        raise NotImplementedError(f"RL_proximal: dataset {args.dataset}")
    return top1.avg


_rl_proximal._ignores_mask = True
_rl_proximal.__name__ = "RL_proximal"
RL_proximal = iterative_unlearn(_rl_proximal)
