"""Golden vectors for the evaluation row: the reference's `trainer.validate` (top-1 in eval mode,
sample weighted) and `evaluation.SVC_MIA` (five attack accuracies) on the tiny BN network and generator-made batches.

    python tests/golden/make_golden_eval.py
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from fixtures import TinyCNN, tiny_batches, tiny_state  # noqa: E402


def main():
    MG.import_reference_classification()
    C = MG.REF + "/Classification"
    stub = types.ModuleType("imagenet")  # only used for ImageNet data dicts
    stub.get_x_y_from_data_dict = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    sys.modules["imagenet"] = stub
    val = MG._load("ref_val", C + "/trainer/val.py")
    svc = MG._load("ref_svc_mia", C + "/evaluation/SVC_MIA.py")
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    from fixtures import eval_loaders
    loaders = eval_loaders()
    ragged = [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in tiny_batches(3, 16, 2300)]
    ragged[-1] = (ragged[-1][0][:5], ragged[-1][1][:5])
    acc = val.validate(ragged, model, nn.CrossEntropyLoss(), SimpleNamespace(imagenet_arch=False, print_freq=50))
    m = svc.SVC_MIA(shadow_train=loaders["shadow_train"], shadow_test=loaders["shadow_test"],
                    target_train=None, target_test=loaders["target_test"], model=model)
    np.savez(os.path.join(HERE, "eval_tinycnn.npz"), validate_top1=float(acc),
             **{"mia_" + k: float(v) for k, v in m.items()})
    print("eval fixtures written:", acc, m)


if __name__ == "__main__":
    main()
