"""Boundary Expanding (reference Classification/unlearn/boundary_ex.py:34-138).

The last Linear layer gets one extra output (a "shadow" class, index num_classes) whose row is freshly
initialised; every forget sample is then trained towards that class.  The expanded layer replaces the old one
in the model, so the flat arena is rebuilt (flat.arena_of notices the new parameters).
"""
import torch
import torch.nn as nn

from .. import utils
from ._steps import run_pass
from .impl import iterative_unlearn


def expand_model(model):
    """Grow the classifier by one "shadow" class (what the reference's boundary_ex.py:34-70 does to the last nn.Linear):
    the module registered last among the model's Linear layers gets a (C+1)-row replacement whose first C rows are the
    trained ones and whose extra row comes from a freshly initialised layer.  That fresh layer is drawn from the CPU
    generator — the same draws as the reference's `nn.Linear(in, C + 1)` constructor on a CPU run — so a run is
    reproducible from torch.manual_seed alone on any device."""
    hit = next(((path, mod) for path, mod in reversed(list(model.named_modules())) if isinstance(mod, nn.Linear)), None)
    if hit is None:
        raise ValueError("No Linear layer found in the model.")
    path, trained = hit
    dev = trained.weight.device
    grown = nn.Linear(trained.in_features, trained.out_features + 1, bias=trained.bias is not None,
                      dtype=trained.weight.dtype)
    with torch.no_grad():
        grown.weight = nn.Parameter(torch.cat([trained.weight.detach(), grown.weight[-1:].to(dev)], dim=0))
        if trained.bias is not None:
            grown.bias = nn.Parameter(torch.cat([trained.bias.detach(), grown.bias[-1:].to(dev)], dim=0))
    parent, _, leaf = path.rpartition(".")
    setattr(model.get_submodule(parent) if parent else model, leaf, grown)


@iterative_unlearn
def boundary_expanding_iter(data_loaders, model, criterion, optimizer, epoch, args, mask=None):
    train_loader = data_loaders["forget"]
    losses, top1 = utils.AverageMeter(), utils.AverageMeter()
    model.train()

    def shadow_labels(image, target):
        return torch.ones_like(target) * args.num_classes

    run_pass(train_loader, model, criterion, optimizer, epoch, args, batch_label_fn=shadow_labels, track=True,
             losses=losses, top1=top1)
    print("train_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg


def boundary_expanding(data_loaders, model, criterion, args, mask=None):
    expand_model(model)
    if mask:
        # the reference multiplies the expanded layer's gradient by the old-shape mask and fails on the shape
        # (boundary_ex.py:10-13 with the (num_classes+1)-row layer); say so instead of a broadcasting error
        fc = [n for n, m in model.named_modules() if isinstance(m, nn.Linear)][-1]
        raise RuntimeError(f"boundary_expanding: mask['{fc}.weight'] has the pre-expansion shape; the reference "
                           "cannot run this method with a saliency mask either")
    return boundary_expanding_iter(data_loaders, model, criterion, args, mask)
