"""Evaluation row: `trainer.validate` and `evaluation.SVC_MIA` of this package against outputs of
the reference's functions on the same model and batches (tests/golden/make_golden_eval.py).  CPU (the functions are
device agnostic; the GPU suite runs them on the device)."""
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from fixtures import TinyCNN, tiny_batches, tiny_state


def _mk(nb, seed):
    return [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in tiny_batches(nb, 16, seed)]


def test_validate_and_svc_mia_match_reference(golden_dir):
    run_eval(golden_dir, "cpu")


def run_eval(golden_dir, device):
    """Shared with the GPU suite (tests/test_next_gpu.py): same checks with the model on `device` (the loaders stay on
    the host, as the reference's DataLoaders do; `validate` / `SVC_MIA` move each batch)."""
    from unlearn_saliency_amd.Classification.evaluation.svc_mia import SVC_MIA
    from unlearn_saliency_amd.Classification.trainer.val import validate
    g = np.load(os.path.join(golden_dir, "eval_tinycnn.npz"))
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.to(device)
    ragged = _mk(3, 2300)
    ragged[-1] = (ragged[-1][0][:5], ragged[-1][1][:5])
    acc = validate(ragged, model, nn.CrossEntropyLoss(), SimpleNamespace(imagenet_arch=False, print_freq=50))
    assert abs(acc - float(g["validate_top1"])) < 1e-4  # sample-weighted top-1 in percent
    from fixtures import eval_loaders
    ld = eval_loaders()
    m = SVC_MIA(shadow_train=ld["shadow_train"], shadow_test=ld["shadow_test"], target_train=None,
                target_test=ld["target_test"], model=model)
    assert len({round(v, 6) for v in m.values()}) >= 3  # the fixture separates the features: not a degenerate attack
    assert set(m) == {"correctness", "confidence", "entropy", "m_entropy", "prob"}
    for k, v in m.items():
        # attack accuracies are ratios of small integer counts: equal unless a feature lands on the SVC's boundary
        assert abs(v - float(g["mia_" + k])) < 1e-6, (k, v, float(g["mia_" + k]))
