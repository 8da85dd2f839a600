"""Gradient ascent on the forget set (reference Classification/unlearn/GA.py:44-153);
GA_l1 adds alpha * ||theta||_1 (GA.py:156-230).  Same fused step as RL."""
from .. import utils
from ._steps import run_pass
from .impl import iterative_unlearn


def _ga_epoch(data_loaders, model, criterion, optimizer, epoch, args, l1_alpha=0.0):
    loader = data_loaders["forget"]
    losses, top1 = utils.AverageMeter(), utils.AverageMeter()
    model.train()
    run_pass(loader, model, criterion, optimizer, epoch, args, loss_sign=-1.0, l1_alpha=l1_alpha, track=True,
             losses=losses, top1=top1, warmup_steps_per_epoch=len(loader))
    print("train_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg


@iterative_unlearn
def GA(data_loaders, model, criterion, optimizer, epoch, args, mask=None):
    return _ga_epoch(data_loaders, model, criterion, optimizer, epoch, args)


@iterative_unlearn
def GA_l1(data_loaders, model, criterion, optimizer, epoch, args, mask=None):
    return _ga_epoch(data_loaders, model, criterion, optimizer, epoch, args, l1_alpha=args.alpha)
