"""Random-label (RL) unlearning — SalUn's Phase B when a saliency mask is passed.

Per epoch (reference Classification/unlearn/RL.py:109-178, cifar10/svhn branch): one pass
over the forget loader with *fresh uniform random labels for every batch*
(`torch.randint(0, num_classes, target.shape)` from the CPU generator, RL.py:125), then one
pass over the retain loader with the true labels; BatchNorm in train mode; returns the
retain-pass top-1.  With a mask, every step is the fused masked-SGD launch (impl.py).
"""
import torch

from .. import utils
from ._steps import run_pass
from .impl import iterative_unlearn


@iterative_unlearn
def RL(data_loaders, model, criterion, optimizer, epoch, args, mask=None):
    forget_loader = data_loaders["forget"]
    retain_loader = data_loaders["retain"]
    if args.dataset not in ("cifar10", "svhn"):
        raise NotImplementedError("RL: only the cifar10/svhn branch (RL.py:109-176) is in scope; the "
                                  "cifar100/TinyImagenet branch relabels a merged dataset instead")
    if epoch < args.warmup:
        # the reference's warm-up here reads an undefined loop variable (RL.py:119-121)
        raise NameError("name 'i' is not defined  [reference RL.py:120: warmup>0 is unusable; keep --warmup 0]")
    losses, top1 = utils.AverageMeter(), utils.AverageMeter()
    model.train()
    loader_len = len(forget_loader) + len(retain_loader)

    def random_labels(target):
        return torch.randint(0, args.num_classes, target.shape)

    run_pass(forget_loader, model, criterion, optimizer, epoch, args, label_fn=random_labels, track=False,
             loader_len=loader_len)
    run_pass(retain_loader, model, criterion, optimizer, epoch, args, track=True, losses=losses, top1=top1,
             loader_len=loader_len)
    return top1.avg
