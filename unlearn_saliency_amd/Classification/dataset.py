"""CIFAR-10 input pipeline for the SalUn classification path (no torchvision needed).

Mirrors the *observable behaviour* of the reference's loader
(Classification/dataset.py:529-705): 45,000 / 5,000 stratified train/val split drawn
with ``np.random.RandomState(seed)``, forget-set marking by negated labels
(``-label-1``) with ``RandomState(seed-1)``, class-wise test-set filtering, and loaders
whose ``.dataset`` exposes ``.data`` (N,32,32,3 uint8), ``.targets`` (ndarray) and
``.transform`` so the entry scripts can slice forget/retain sets exactly as the
reference scripts do (main_random.py:50-110).

Two batch paths:

* host path (default) — per-sample RandomCrop(32, padding=4) + RandomHorizontalFlip +
  ToTensor consuming the global torch RNG in torchvision's order (two ``randint`` then
  one ``rand`` per sample), shuffling like ``torch.utils.data.RandomSampler``; this is
  the reference's K0 pipeline kept for RNG-stream compatibility;
* device path (``device_resident=True``) — the whole uint8 set lives in HBM
  (45,000 x 3,072 B = 138 MB) and ``salun_image_batch`` assembles the fp32 batch
  (gather + crop + flip + /255) in one kernel, removing the single-threaded host loop
  that otherwise bounds steps/s.
"""
from __future__ import annotations

import copy
import os
import pickle
from typing import Optional

import numpy as np
import torch

from .. import rng as salun_rng

TRAIN_TRANSFORM = "train"  # RandomCrop(32, 4) + RandomHorizontalFlip + ToTensor
TEST_TRANSFORM = "test"    # ToTensor only

SYNTHETIC_SEED = 2


class ArrayDataset:
    """In-memory image set with torchvision-CIFAR10-like attributes."""

    def __init__(self, data: np.ndarray, targets: np.ndarray, transform: str = TEST_TRANSFORM, train: bool = True):
        assert data.dtype == np.uint8 and data.ndim == 4 and data.shape[-1] == 3
        self.data = data
        self.targets = np.asarray(targets)
        self.transform = transform
        self.train = train

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        img = self.data[i]
        if self.transform == TRAIN_TRANSFORM:
            # torchvision order: RandomCrop.get_params (randint i, randint j), then flip (rand < 0.5)
            dy = int(torch.randint(0, 9, size=(1,)).item())
            dx = int(torch.randint(0, 9, size=(1,)).item())
            flip = bool(torch.rand(1) < 0.5)
            padded = np.zeros((40, 40, 3), np.uint8)
            padded[4:36, 4:36] = img
            img = padded[dy:dy + 32, dx:dx + 32]
            if flip:
                img = img[:, ::-1]
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(torch.float32).div(255)
        return x, int(self.targets[i])


class BatchLoader:
    """The subset of torch.utils.data.DataLoader the unlearning loops rely on:
    iteration over (image[B,3,32,32] fp32, target[B] int64), ``len()``, ``.dataset``,
    ``.batch_size``.  `shuffle=True` draws one permutation per epoch the way
    RandomSampler does (a 64-bit seed from the global generator, then randperm)."""

    def __init__(self, dataset: ArrayDataset, batch_size: int, shuffle: bool, device_resident: bool = False,
                 device: Optional[torch.device] = None, rank: int = 0, world_size: int = 1):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.device_resident = device_resident
        self.device = device
        self.rank, self.world_size = rank, world_size
        self._dev_cache = None  # (id(data), data_ptr-holder, targets)
        # (lo, hi, b) of the batch just yielded: this rank holds samples [lo, hi) of a global batch of b — what the
        # callers need to draw per-batch randomness for the GLOBAL batch (then slice) and to count-weight the loss
        self.last_shard = None

    def _slice(self, b: int):
        if self.world_size <= 1:
            return 0, b
        return self.rank * b // self.world_size, (self.rank + 1) * b // self.world_size

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _order(self) -> torch.Tensor:
        n = len(self.dataset)
        # DataLoader draws a base seed when the iterator is created, RandomSampler another when started
        torch.empty((), dtype=torch.int64).random_()
        if not self.shuffle:
            return torch.arange(n)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g)

    def _resident(self):
        ds = self.dataset
        key = (id(ds.data), ds.data.shape, id(ds.targets))
        if self._dev_cache is None or self._dev_cache[0] != key:
            dev = self.device or torch.device("cuda", torch.cuda.current_device())
            data = torch.from_numpy(np.ascontiguousarray(ds.data)).to(dev)
            targets = torch.from_numpy(np.asarray(ds.targets).astype(np.int64)).to(dev)
            self._dev_cache = (key, data, targets)
        return self._dev_cache[1], self._dev_cache[2]

    def __iter__(self):
        order = self._order()
        n, bs = len(self.dataset), self.batch_size
        if self.device_resident:
            from .. import ops
            data, targets = self._resident()
            order = order.to(data.device)
            train = self.dataset.transform == TRAIN_TRANSFORM
            for s in range(0, n, bs):
                idx = order[s:s + bs]
                b = idx.numel()
                lo, hi = self._slice(b)  # contiguous, balanced shard of the global batch
                crop = flip = None
                if train:
                    # augmentation is drawn for the GLOBAL batch on every rank (identically seeded generators), then
                    # sliced: ranks stay in lock-step whatever their shard sizes, and the global batch sees b
                    # independent draws, not the same b/ws draws on every rank
                    crop = torch.randint(0, 9, (b, 2), device=data.device, dtype=torch.int32)[lo:hi].contiguous()
                    flip = (torch.rand(b, device=data.device) < 0.5).to(torch.uint8)[lo:hi].contiguous()
                idx = idx[lo:hi].contiguous()
                self.last_shard = (lo, hi, b)
                yield ops.image_batch(data, idx, crop, flip, pad=4), targets[idx]
        else:
            order = order.tolist()
            for s in range(0, n, bs):
                idx = order[s:s + bs]
                lo, hi = self._slice(len(idx))
                self.last_shard = (lo, hi, len(idx))
                idx = idx[lo:hi]
                items = [self.dataset[i] for i in idx]
                x = torch.stack([it[0] for it in items]) if items else torch.empty(0, 3, 32, 32)
                y = torch.tensor([it[1] for it in items], dtype=torch.int64)
                yield x, y


# ------------------------------------------------------------------------- sources
def _load_cifar10_files(data_dir: str):
    """Standard `cifar-10-batches-py` pickles (what torchvision's CIFAR10 reads; no download here)."""
    base = os.path.join(data_dir, "cifar-10-batches-py")

    def read(names):
        xs, ys = [], []
        for nm in names:
            with open(os.path.join(base, nm), "rb") as f:
                d = pickle.load(f, encoding="latin1")
            xs.append(np.asarray(d["data"], dtype=np.uint8))
            ys.extend(d["labels"] if "labels" in d else d["fine_labels"])
        x = np.vstack(xs).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
        return np.ascontiguousarray(x), np.asarray(ys, dtype=np.int64)

    return read([f"data_batch_{i}" for i in range(1, 6)]), read(["test_batch"])


def synthetic_cifar10(n_train: int = 50_000, n_test: int = 10_000, seed: int = SYNTHETIC_SEED):
    """CIFAR-shaped stand-in: uint8 pixels ~ U{0..255} from the counter-based
    generator, balanced labels i % 10.  After the 10 % validation split: 45,000 train, 4,500/class."""
    xtr = salun_rng.u8(n_train * 3072, seed).reshape(n_train, 32, 32, 3)
    xte = salun_rng.u8(n_test * 3072, seed + 1_000_003).reshape(n_test, 32, 32, 3)
    ytr = (np.arange(n_train) % 10).astype(np.int64)
    yte = (np.arange(n_test) % 10).astype(np.int64)
    return (xtr, ytr), (xte, yte)


def have_cifar10(data_dir: str) -> bool:
    return os.path.exists(os.path.join(data_dir, "cifar-10-batches-py", "data_batch_1"))


# ---------------------------------------------------------------- forget marking
def replace_indexes(dataset, indexes, seed=0, only_mark: bool = False):
    """only_mark: flag forget samples by label -> -label-1 (the -1 keeps class 0 markable);
    otherwise overwrite them with random other samples (reference dataset.py:648-671)."""
    indexes = np.asarray(indexes)
    if only_mark:
        dataset.targets[indexes] = -dataset.targets[indexes] - 1
        return
    rng = np.random.RandomState(seed)
    pool = list(set(range(len(dataset))) - set(indexes.tolist()))
    new_indexes = rng.choice(pool, size=len(indexes))
    dataset.data[indexes] = dataset.data[new_indexes]
    dataset.targets[indexes] = dataset.targets[new_indexes]


def replace_class(dataset, class_to_replace: int, num_indexes_to_replace: Optional[int] = None, seed: int = 0,
                  only_mark: bool = False):
    """class -1 = "any class" (random-data forgetting); reference dataset.py:674-705."""
    targets = np.asarray(dataset.targets)
    if class_to_replace == -1:
        indexes = np.flatnonzero(np.ones_like(targets))
    else:
        indexes = np.flatnonzero(targets == class_to_replace)
    if num_indexes_to_replace is not None:
        if num_indexes_to_replace > len(indexes):
            raise AssertionError(f"Want to replace {num_indexes_to_replace} indexes but only {len(indexes)} "
                                 "samples in dataset")
        rng = np.random.RandomState(seed)
        indexes = rng.choice(indexes, size=num_indexes_to_replace, replace=False)
        print(f"Replacing indexes {indexes}")
    replace_indexes(dataset, indexes, seed, only_mark)


def cifar10_dataloaders(batch_size=128, data_dir="datasets/cifar10", num_workers=2, random_to_replace=None,
                        class_to_replace: Optional[int] = None, num_indexes_to_replace=None,
                        indexes_to_replace=None, seed: int = 1, only_mark: bool = False, shuffle=True,
                        no_aug=False, synthetic: bool = False, device_resident: bool = False):
    """(train_loader, val_loader, test_loader), same splits as reference dataset.py:529-645."""
    if synthetic or not have_cifar10(data_dir):
        if not synthetic:
            print(f"CIFAR-10 files not found under {data_dir!r}: using the synthetic CIFAR-shaped set")
        (xtr, ytr), (xte, yte) = synthetic_cifar10()
    else:
        (xtr, ytr), (xte, yte) = _load_cifar10_files(data_dir)
    print("Dataset information: CIFAR-10\t 45000 images for training \t 5000 images for validation\t")
    print("10000 images for testing\t no normalize applied in data_transform")

    rng = np.random.RandomState(seed)
    valid_idx = []
    for c in range(int(ytr.max()) + 1):
        class_idx = np.where(ytr == c)[0]
        valid_idx.append(rng.choice(class_idx, int(0.1 * len(class_idx)), replace=False))
    valid_idx = np.hstack(valid_idx)
    train_idx = np.asarray(sorted(set(range(len(xtr))) - set(valid_idx.tolist())), dtype=np.int64)

    train_tf = TEST_TRANSFORM if no_aug else TRAIN_TRANSFORM
    train_set = ArrayDataset(xtr[train_idx], ytr[train_idx].copy(), train_tf, train=True)
    valid_set = ArrayDataset(xtr[valid_idx], ytr[valid_idx].copy(), train_tf, train=True)
    test_set = ArrayDataset(xte, yte.copy(), TEST_TRANSFORM, train=False)

    if class_to_replace is not None and indexes_to_replace is not None:
        raise ValueError("Only one of `class_to_replace` and `indexes_to_replace` can be specified")
    if class_to_replace is not None:
        replace_class(train_set, class_to_replace, num_indexes_to_replace=num_indexes_to_replace, seed=seed - 1,
                      only_mark=only_mark)
        if num_indexes_to_replace is None or num_indexes_to_replace == 4500:
            keep = test_set.targets != class_to_replace
            test_set.data, test_set.targets = test_set.data[keep], test_set.targets[keep]
    if indexes_to_replace is not None:
        replace_indexes(train_set, indexes_to_replace, seed=seed - 1, only_mark=only_mark)

    mk = lambda ds, sh: BatchLoader(ds, batch_size, sh, device_resident=device_resident)
    return mk(train_set, True), mk(valid_set, False), mk(test_set, False)


def split_marked(marked_dataset: ArrayDataset):
    """forget (targets < 0, restored to -t-1) / retain (targets >= 0) copies of a marked set —
    the slicing the reference scripts do inline (generate_mask.py:148-164, main_random.py:77-93)."""
    forget = copy.deepcopy(marked_dataset)
    sel = forget.targets < 0
    forget.data, forget.targets = forget.data[sel], -forget.targets[sel] - 1
    retain = copy.deepcopy(marked_dataset)
    sel = retain.targets >= 0
    retain.data, retain.targets = retain.data[sel], retain.targets[sel]
    return forget, retain
