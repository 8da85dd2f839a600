"""Plumbing the drop-in must reproduce (reference Classification/utils.py): seeds, meters,
top-k accuracy, checkpoint naming, and `setup_model_dataset` (CIFAR-10 / ResNet-18 path)."""
from __future__ import annotations

import os
import random
import shutil

import numpy as np
import torch

from .dataset import TEST_TRANSFORM, TRAIN_TRANSFORM, cifar10_dataloaders
from .models import model_dict
from .models.resnet_cifar import CIFAR10_MEAN, CIFAR10_STD, NormalizeByChannelMeanStd

__all__ = ["setup_model_dataset", "setup_seed", "AverageMeter", "accuracy", "save_checkpoint", "load_checkpoint",
           "NormalizeByChannelMeanStd", "dataset_convert_to_test", "dataset_convert_to_train", "warmup_lr"]


def setup_seed(seed: int) -> None:
    """All four generators + deterministic conv algorithms (reference utils.py:288-294)."""
    print("setup random seed = {}".format(seed))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    torch.backends.cudnn.deterministic = True


class AverageMeter:
    """Running sample-weighted mean: .val (last), .avg, .sum, .count."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """precision@k in percent, one 1-element tensor per k (reference utils.py:321-334)."""
    kmax = max(topk)
    pred = output.topk(kmax, dim=1, largest=True, sorted=True).indices.t()  # (kmax, B)
    hit = pred.eq(target.view(1, -1).expand_as(pred))
    n = target.size(0)
    return [hit[:k].reshape(-1).float().sum(0).mul_(100.0 / n) for k in topk]


def warmup_lr(epoch, step, optimizer, one_epoch_step, args):
    overall = args.warmup * one_epoch_step
    lr = min(args.lr * (step + epoch * one_epoch_step) / overall, args.lr)
    for group in optimizer.param_groups:
        group["lr"] = lr


def save_checkpoint(state, is_SA_best, save_path, pruning, filename="checkpoint.pth.tar"):
    """{save_path}/{pruning}{filename} — e.g. RLcheckpoint.pth.tar (reference utils.py:44-52)."""
    path = os.path.join(save_path, str(pruning) + filename)
    torch.save(state, path)
    if is_SA_best:
        shutil.copyfile(path, os.path.join(save_path, str(pruning) + "model_SA_best.pth.tar"))


def load_checkpoint(device, save_path, pruning, filename="checkpoint.pth.tar"):
    path = os.path.join(save_path, str(pruning) + filename)
    if not os.path.exists(path):
        print("Checkpoint not found! path:{}".format(path))
        return None
    print("Load checkpoint from:{}".format(path))
    return torch.load(path, map_location=device, weights_only=False)


def _innermost(dataset):
    while hasattr(dataset, "dataset"):
        dataset = dataset.dataset
    return dataset


def dataset_convert_to_train(dataset):
    ds = _innermost(dataset)
    ds.transform = TRAIN_TRANSFORM
    ds.train = False


def dataset_convert_to_test(dataset, args=None):
    """Switch a (possibly wrapped) dataset to the evaluation transform (reference utils.py:97-109)."""
    ds = _innermost(dataset)
    ds.transform = TEST_TRANSFORM
    ds.train = False


def setup_model_dataset(args):
    """-> (model, train_full_loader, val_loader, test_loader, marked_loader), CIFAR-10 branch of
    reference utils.py:112-146: a full loader, a second set of loaders with the forget samples
    *marked* (labels negated, seed = args.seed so RandomState(seed-1) picks them), the model built
    under `train_seed` and the normalisation layer installed on it."""
    if args.dataset != "cifar10":
        raise NotImplementedError(f"dataset {args.dataset!r} is outside the hot-path scope (SURVEY.md §2 C6); "
                                  "only cifar10 is in the benchmark configs")
    synthetic = bool(getattr(args, "synthetic", False))
    resident = bool(getattr(args, "device_loader", False))
    common = dict(batch_size=args.batch_size, data_dir=args.data, num_workers=args.workers, synthetic=synthetic,
                  device_resident=resident)
    train_full_loader, val_loader, _ = cifar10_dataloaders(**common)
    marked_loader, _, test_loader = cifar10_dataloaders(
        class_to_replace=args.class_to_replace, num_indexes_to_replace=args.num_indexes_to_replace,
        indexes_to_replace=args.indexes_to_replace, seed=args.seed, only_mark=True, shuffle=True,
        no_aug=args.no_aug, **common)
    if args.train_seed is None:
        args.train_seed = args.seed
    setup_seed(args.train_seed)
    model = model_dict[args.arch](num_classes=10, imagenet=bool(args.imagenet_arch))
    setup_seed(args.train_seed)
    model.normalize = NormalizeByChannelMeanStd(CIFAR10_MEAN, CIFAR10_STD)
    return model, train_full_loader, val_loader, test_loader, marked_loader
