"""The `unlearn` plugin registry (reference Classification/unlearn/__init__.py:18-61).

`get_unlearn_method(name)` returns a callable
``method(data_loaders, model, criterion, args, mask=None) -> None`` that mutates `model`
in place; unknown names raise NotImplementedError exactly like the reference.  All 17
registry names are kept so command lines stay drop-in; the SalUn hot path (RL with a
mask) and the baselines that share its fused step (GA, GA_l1, FT, FT_l1, raw) are
implemented, the remaining baselines are registered but raise with a scope note
(they are the "next" rows, not part of the accelerated path).
"""
from .FT import FT, FT_l1
from .GA import GA, GA_l1
from .impl import (FusedMaskedSGD, iterative_unlearn, load_unlearn_checkpoint, save_unlearn_checkpoint)
from .RL import RL


def raw(data_loaders, model, criterion, args, mask=None):
    """No unlearning: evaluate the original model."""
    return None


def _out_of_scope(name, why):
    def method(data_loaders, model, criterion, args, mask=None):
        raise NotImplementedError(f"Unlearn method {name} is registered for CLI compatibility but is outside the "
                                  f"accelerated hot path of this build ({why}); see SURVEY.md §8 (f)")
    method.__name__ = name
    return method


_REGISTRY = {
    "raw": raw, "RL": RL, "GA": GA, "FT": FT, "FT_l1": FT_l1, "GA_l1": GA_l1,
    "retrain": _out_of_scope("retrain", "re-training from scratch is pre-training, not unlearning arithmetic"),
    "fisher": _out_of_scope("fisher", "Fisher-forgetting baseline"),
    "fisher_new": _out_of_scope("fisher_new", "Fisher-forgetting baseline"),
    "wfisher": _out_of_scope("wfisher", "influence-unlearning baseline"),
    "FT_prune": _out_of_scope("FT_prune", "pruning baseline"),
    "FT_prune_bi": _out_of_scope("FT_prune_bi", "pruning baseline"),
    "GA_prune": _out_of_scope("GA_prune", "pruning baseline"),
    "GA_prune_bi": _out_of_scope("GA_prune_bi", "pruning baseline"),
    "boundary_expanding": _out_of_scope("boundary_expanding", "boundary-unlearning baseline, F1 next"),
    "boundary_shrink": _out_of_scope("boundary_shrink", "boundary-unlearning baseline, F1 next"),
    "RL_proximal": _out_of_scope("RL_proximal", "proximal variant, F2 next"),
}


def get_unlearn_method(name):
    """method usage:  function(data_loaders, model, criterion, args, mask=None)"""
    try:
        return _REGISTRY[name]
    except KeyError:
        raise NotImplementedError(f"Unlearn method {name} not implemented!") from None
