"""BASELINE.json configs[2] — class-wise forgetting: the forget-set marking and the class-wise test-set filter of
`Classification/dataset.py` against what the REFERENCE's `cifar10_dataloaders` / `replace_class` / `replace_indexes`
(reference dataset.py:529-705) produced on the same synthetic arrays (`tests/golden/classwise.npz`, written by
`tests/golden/make_golden_classwise.py`).  Integer / byte work: everything is compared exactly."""
import hashlib
import os
from types import SimpleNamespace

import numpy as np
import pytest

from unlearn_saliency_amd.Classification import dataset as D

CASES = {"class0_all": (0, None), "class0_4500": (0, 4500), "class0_2000": (0, 2000), "class3_all": (3, None),
         "random_4500": (-1, 4500)}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "classwise.npz"))


@pytest.mark.parametrize("tag", list(CASES))
def test_marking_and_test_filter_equal_the_reference(golden, tag, capsys):
    cls, num = CASES[tag]
    tr, va, te = D.cifar10_dataloaders(batch_size=256, synthetic=True, class_to_replace=cls,
                                       num_indexes_to_replace=num, seed=2, only_mark=True, shuffle=True)
    capsys.readouterr()
    want = golden[f"{tag}__marked_targets"].astype(np.int64)
    assert np.array_equal(np.asarray(tr.dataset.targets), want)
    assert np.array_equal(np.asarray(te.dataset.targets), golden[f"{tag}__test_targets"].astype(np.int64))
    assert np.array_equal(np.asarray(va.dataset.targets), golden[f"{tag}__valid_targets"].astype(np.int64))
    assert [sha(tr.dataset.data), sha(te.dataset.data), sha(va.dataset.data)] == list(golden[f"{tag}__sha"])
    forget, retain = D.split_marked(tr.dataset)
    n_forget = 4500 if num is None else num
    assert len(forget) == n_forget and len(retain) == 45000 - n_forget
    assert forget.targets.min() >= 0 and retain.targets.min() >= 0
    if cls >= 0:
        assert set(forget.targets.tolist()) == {cls}
        # the whole class is removed from the test set only when the whole class is forgotten (dataset.py:606-608)
        whole = num is None or num == 4500
        assert (cls not in set(te.dataset.targets.tolist())) == whole
        assert len(te.dataset) == (9000 if whole else 10000)
        assert (cls in set(retain.targets.tolist())) == (not whole)
    else:
        assert len(te.dataset) == 10000 and len(set(forget.targets.tolist())) == 10


def test_setup_model_dataset_passes_the_class_through(golden, capsys):
    """`--class_to_replace 0 --seed 2` through the argument parser and `utils.setup_model_dataset` (reference
    utils.py:112-146) lands on the same marked set."""
    from unlearn_saliency_amd.Classification import arg_parser, utils
    args = arg_parser.parse_args(["--class_to_replace", "0", "--seed", "2", "--synthetic", "--batch_size", "256",
                                  "--save_dir", "/tmp/none"])
    assert args.num_indexes_to_replace is None
    model, full, val, test, marked = utils.setup_model_dataset(args)
    capsys.readouterr()
    assert np.array_equal(np.asarray(marked.dataset.targets), golden["class0_all__marked_targets"].astype(np.int64))
    assert len(test.dataset) == 9000 and len(full.dataset) == 45000
    assert (np.asarray(full.dataset.targets) >= 0).all()  # the full loader is never marked


@pytest.mark.parametrize("tag,cls,num", [("small_overwrite_class1", 1, None), ("small_overwrite_random", -1, 150)])
def test_overwrite_branch_writes_what_the_reference_wrote(golden, tag, cls, num, capsys):
    """only_mark=False (outside the unlearning path).  The reference overwrites data / targets and then raises
    AttributeError from a misplaced `else:` (dataset.py:655-662); the arrays it leaves behind are the contract,
    the exception is not reproduced."""
    (xs, ys), _ = D.synthetic_cifar10(n_train=2000, n_test=400, seed=77)
    ds = D.ArrayDataset(xs.copy(), ys.copy())
    D.replace_class(ds, cls, num_indexes_to_replace=num, seed=1, only_mark=False)
    capsys.readouterr()
    assert str(golden[f"{tag}__raised"]) == "AttributeError"
    assert np.array_equal(ds.targets, golden[f"{tag}__targets"].astype(np.int64))
    assert sha(ds.data) == str(golden[f"{tag}__sha"][0])


def test_both_selectors_are_refused():
    with pytest.raises(ValueError):
        D.cifar10_dataloaders(synthetic=True, class_to_replace=0, indexes_to_replace=[1, 2], seed=2)
    with pytest.raises(AssertionError):
        (xs, ys), _ = D.synthetic_cifar10(n_train=200, n_test=10)
        D.replace_class(D.ArrayDataset(xs, ys), 0, num_indexes_to_replace=21, seed=1, only_mark=True)
