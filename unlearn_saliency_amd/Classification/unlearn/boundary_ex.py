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
    """Replace the last nn.Linear by one with out_features + 1; old rows copied (boundary_ex.py:34-70).
    The new layer is initialised from the CPU generator and then moved, so a run is reproducible from
    torch.manual_seed alone on any device."""
    last_fc_name, last_fc_layer = None, None
    for name, module in model.named_modules():
        if isinstance(module, nn.Linear):
            last_fc_name, last_fc_layer = name, module
    if last_fc_name is None:
        raise ValueError("No Linear layer found in the model.")
    num_classes = last_fc_layer.out_features
    bias = last_fc_layer.bias is not None
    new_fc = nn.Linear(last_fc_layer.in_features, num_classes + 1, bias=bias, dtype=last_fc_layer.weight.dtype)
    new_fc = new_fc.to(last_fc_layer.weight.device)
    with torch.no_grad():
        new_fc.weight[:-1] = last_fc_layer.weight
        if bias:
            new_fc.bias[:-1] = last_fc_layer.bias
    parts = last_fc_name.split(".")
    owner = model
    for part in parts[:-1]:
        owner = getattr(owner, part)
    setattr(owner, parts[-1], new_fc)


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
