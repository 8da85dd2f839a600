"""Fine-tuning on the retain set (reference Classification/unlearn/FT.py:44-180).  FT_l1 adds
the decaying l1 penalty  alpha * (1 - epoch / (unlearn_epochs - no_l1_epochs))  (FT.py:131-136)."""
from .. import utils
from ._steps import run_pass
from .impl import iterative_unlearn


def FT_iter(data_loaders, model, criterion, optimizer, epoch, args, mask=None, with_l1=False):
    loader = data_loaders["retain"]
    losses, top1 = utils.AverageMeter(), utils.AverageMeter()
    model.train()
    alpha = 0.0
    if with_l1:
        span = args.unlearn_epochs - args.no_l1_epochs
        alpha = args.alpha * (1 - epoch / span) if epoch < span else 0.0
    run_pass(loader, model, criterion, optimizer, epoch, args, l1_alpha=alpha, track=True, losses=losses, top1=top1,
             warmup_steps_per_epoch=len(loader))
    print("train_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg


@iterative_unlearn
def FT(data_loaders, model, criterion, optimizer, epoch, args, mask=None):
    return FT_iter(data_loaders, model, criterion, optimizer, epoch, args, mask)


@iterative_unlearn
def FT_l1(data_loaders, model, criterion, optimizer, epoch, args, mask=None):
    return FT_iter(data_loaders, model, criterion, optimizer, epoch, args, mask, with_l1=True)
