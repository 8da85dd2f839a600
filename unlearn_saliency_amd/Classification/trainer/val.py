"""`validate`: eval-mode top-1 over a loader (reference Classification/trainer/val.py:6-72).
Unlearning accuracy is 100 - validate(forget_loader)."""
import torch

from .. import utils


def _device_of(model):
    return next(model.parameters()).device


def validate(val_loader, model, criterion, args):
    losses, top1 = utils.AverageMeter(), utils.AverageMeter()
    model.eval()
    dev = _device_of(model)
    print_freq = getattr(args, "print_freq", 50)
    # per-batch host syncs are avoided: sums are kept on the device and read once
    loss_sum = torch.zeros((), device=dev, dtype=torch.float64)
    hit_sum = torch.zeros((), device=dev, dtype=torch.float64)
    count = 0
    for i, (image, target) in enumerate(val_loader):
        image, target = image.to(dev, non_blocking=True), target.to(dev, non_blocking=True)
        with torch.no_grad():
            output = model(image).float()
            loss = criterion(output, target).float()
        n = image.size(0)
        loss_sum += loss.double() * n
        hit_sum += (output.argmax(dim=1) == target).sum().double()
        count += n
        if i % print_freq == 0 and getattr(args, "verbose_eval", False):
            print("Test: [{0}/{1}]".format(i, len(val_loader)))
    if count:
        losses.update(float(loss_sum.item()) / count, count)
        top1.update(float(hit_sum.item()) * 100.0 / count, count)
    print("valid_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg
