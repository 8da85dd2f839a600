"""ringpack.py's registry on the host (no GPU): which weights are registered, that a re-homed parameter (FlatArena
built after registration) is found at its new address, and that dead or foreign tensors are never served."""
import gc

import torch
import torch.nn as nn


def test_registry_follows_rehomed_parameters_and_forgets_dead_ones():
    from unlearn_saliency_amd import ringpack
    a = nn.Conv2d(16, 32, 3, padding=1)
    b = nn.Conv2d(32, 12, 3, padding=1)     # K = 12: backward-data would reduce over 12 channels -> not eligible
    g = ringpack.register([a.weight, b.weight, a.bias])
    assert g is not None and len(g.entries) == 1
    e = ringpack._lookup(a.weight)
    assert e is not None and e.ref() is a.weight
    assert ringpack.register([a.weight]) is None          # already registered
    old = a.weight.data_ptr()
    with torch.no_grad():
        a.weight.data = a.weight.data.clone()              # what FlatArena does: the parameter object moves
    assert a.weight.data_ptr() != old
    e2 = ringpack._lookup(a.weight)
    assert e2 is e and e.ptr == a.weight.data_ptr() and e.key is None
    assert ringpack._lookup(torch.zeros(32, 16, 3, 3)) is None
    assert ringpack.images(a.weight) is None               # host tensor: no image (the kernels have no CPU path)
    del a, e, e2, g
    gc.collect()
    ringpack._rebuild()
    assert all(x.ref() is not None for x in ringpack._all)


def test_eligibility():
    from unlearn_saliency_amd import ringpack
    assert ringpack.eligible(64, 64, 3, 1, 1)
    assert not ringpack.eligible(64, 3, 3, 1, 1)           # RGB stem: 3 reduction channels
    assert not ringpack.eligible(64, 64, 3, 2, 1) and not ringpack.eligible(64, 64, 1, 1, 0)
