"""Golden vectors for BASELINE.json configs[2] — class-wise forgetting.

    python tests/golden/make_golden_classwise.py

Calls the REFERENCE's `dataset.cifar10_dataloaders` (Classification/dataset.py:529-645), which in turn runs its
`replace_class` / `replace_indexes` (dataset.py:648-705) and the class-wise test-set filter (dataset.py:599-609), on
the CIFAR-shaped synthetic arrays of SURVEY.md §8 D1 handed in through a stand-in for `torchvision.datasets.CIFAR10`
(the only thing stubbed: where the pixels come from).  Arguments are the ones `utils.setup_model_dataset`
(utils.py:120-131) passes for `--seed 2`: `seed=2, only_mark=True, shuffle=True`.

Stored per case (data only): the marked label vector of the 45,000-sample train set (forget samples carry
`-label-1`), the test-set labels after the filter, the validation labels, and SHA-256 digests of the three pixel
arrays (pins which rows were selected and in which order).  `only_mark=False` (forget samples overwritten by random
other samples — used by the reference's retraining baselines, not by the unlearning path) is pinned on a 2,000-sample
set: the arrays as the reference left them and the exception it ends with.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (import shims for the reference)

from unlearn_saliency_amd.Classification.dataset import synthetic_cifar10  # noqa: E402  (input arrays only)

CASES = {
    # tag: (class_to_replace, num_indexes_to_replace)
    "class0_all": (0, None),        # configs[2]: `--class_to_replace 0`
    "class0_4500": (0, 4500),       # same set through rng.choice, test set filtered as well
    "class0_2000": (0, 2000),       # part of a class: test set keeps the class
    "class3_all": (3, None),
    "random_4500": (-1, 4500),      # configs[1]: `--num_indexes_to_replace 4500` (parse default class -1)
}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    MG.import_reference_classification()
    import dataset as ref_dataset  # the reference's Classification/dataset.py

    arrays = {}

    class FakeCIFAR10:
        """torchvision.datasets.CIFAR10 stand-in: .data (N,32,32,3) uint8, .targets list, len()."""

        def __init__(self, root, train=True, transform=None, download=False):
            x, y = arrays["train" if train else "test"]
            self.data, self.targets, self.transform, self.train = x.copy(), list(y), transform, train

        def __len__(self):
            return len(self.data)

        def __getitem__(self, i):
            return self.data[i], int(self.targets[i])

    ref_dataset.CIFAR10 = FakeCIFAR10
    out = {}
    (xtr, ytr), (xte, yte) = synthetic_cifar10()
    arrays["train"], arrays["test"] = (xtr, ytr), (xte, yte)
    for tag, (cls, num) in CASES.items():
        tr, va, te = ref_dataset.cifar10_dataloaders(batch_size=256, data_dir="/nonexistent", num_workers=0,
                                                     class_to_replace=cls, num_indexes_to_replace=num, seed=2,
                                                     only_mark=True, shuffle=True, no_aug=False)
        t = np.asarray(tr.dataset.targets)
        assert t.min() >= -10 and t.max() <= 9
        out[f"{tag}__marked_targets"] = t.astype(np.int8)
        out[f"{tag}__test_targets"] = np.asarray(te.dataset.targets).astype(np.int8)
        out[f"{tag}__valid_targets"] = np.asarray(va.dataset.targets).astype(np.int8)
        out[f"{tag}__sha"] = np.array([sha(tr.dataset.data), sha(te.dataset.data), sha(va.dataset.data)])
        print(tag, "forget", int((t < 0).sum()), "test", len(te.dataset), "classes marked",
              sorted(set((-t[t < 0] - 1).tolist())))
    # the overwrite branch (only_mark=False; the reference's retraining baselines, not the unlearning path): on a
    # CIFAR10-shaped dataset the reference's `replace_indexes` writes data / targets and then dies in its
    # `try ... except ... else: dataset._labels[...]` clause (dataset.py:655-662) — recorded, with the arrays as it
    # left them, on a 2,000-sample set
    (xs, ys), (xt, yt) = synthetic_cifar10(n_train=2000, n_test=400, seed=77)
    for tag, (cls, num) in {"small_overwrite_class1": (1, None), "small_overwrite_random": (-1, 150)}.items():
        ds = FakeCIFAR10.__new__(FakeCIFAR10)
        ds.data, ds.targets = xs.copy(), ys.copy()
        try:
            ref_dataset.replace_class(ds, cls, num_indexes_to_replace=num, seed=1, only_mark=False)
            raised = ""
        except AttributeError as e:
            raised = type(e).__name__
        out[f"{tag}__raised"] = np.array(raised)
        out[f"{tag}__targets"] = np.asarray(ds.targets).astype(np.int8)
        out[f"{tag}__sha"] = np.array([sha(ds.data)])
        print(tag, "raised:", raised or "nothing")
    np.savez_compressed(os.path.join(HERE, "classwise.npz"), **out)
    print("wrote classwise.npz", os.path.getsize(os.path.join(HERE, "classwise.npz")), "bytes")


if __name__ == "__main__":
    main()
