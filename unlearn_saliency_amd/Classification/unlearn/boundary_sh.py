"""Boundary Shrink (reference Classification/unlearn/boundary_sh.py:37-141).

Per forget batch: an FGSM step of size 0.1 on a frozen copy of the original model finds each sample's nearest
*other* class (the argmax of the frozen model on the perturbed image); the model is then trained on the clean
image towards that neighbour label.  Accuracy is reported against the true labels.  The step itself is the same
fused masked-SGD launch as RL — with a SalUn mask the update is confined to the salient weights.
"""
import copy

import torch

from .. import utils
from ._steps import run_pass
from .impl import iterative_unlearn

BOUND = 0.1  # "hard coding in the paper" (boundary_sh.py:68)


def discretize(x):
    return torch.round(x * 255) / 255


def FGSM_perturb(x, y, model=None, bound=None, criterion=None):
    """x_adv = discretize(clamp(x + bound * sign(d loss / d x), 0, 1))  (boundary_sh.py:39-52)."""
    device = next(model.parameters()).device
    model.zero_grad()
    x_adv = x.detach().clone().to(device).requires_grad_(True)
    loss = criterion(model(x_adv), y)
    grad, = torch.autograd.grad(loss, x_adv)  # only the input gradient is needed: no weight-gradient kernels
    x_adv = x_adv.detach() + grad.sign() * bound
    return discretize(torch.clamp(x_adv, 0.0, 1.0)).detach()


@iterative_unlearn
def boundary_shrink_iter(data_loaders, model, criterion, optimizer, epoch, args, mask=None, test_model=None):
    assert test_model is not None
    train_loader = data_loaders["forget"]
    losses, top1 = utils.AverageMeter(), utils.AverageMeter()
    model.train()

    def neighbour_labels(image, target):
        test_model.eval()
        image_adv = FGSM_perturb(image, target, model=test_model, bound=BOUND, criterion=criterion)
        with torch.no_grad():
            return torch.argmax(test_model(image_adv), dim=1)

    run_pass(train_loader, model, criterion, optimizer, epoch, args, batch_label_fn=neighbour_labels, track=True,
             losses=losses, top1=top1, warmup_steps_per_epoch=len(train_loader))
    print("train_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg


def boundary_shrink(data_loaders, model, criterion, args, mask=None):
    device = next(model.parameters()).device
    test_model = copy.deepcopy(model).to(device)
    for p in test_model.parameters():  # frozen: FGSM differentiates w.r.t. the image only
        p.requires_grad_(False)
    return boundary_shrink_iter(data_loaders, model, criterion, args, mask, test_model=test_model)
