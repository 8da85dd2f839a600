"""The `unlearn` plugin registry (reference Classification/unlearn/__init__.py:18-61).

`get_unlearn_method(name)` returns a callable
``method(data_loaders, model, criterion, args, mask=None) -> None`` that mutates `model`
in place; unknown names raise NotImplementedError exactly like the reference.  All 17
registry names are kept so command lines stay drop-in; the SalUn hot path (RL with a
mask), the baselines that share its fused step (GA, GA_l1, FT, FT_l1, raw, boundary_shrink,
boundary_expanding — SURVEY.md §8 F1) and the proximal variant (RL_proximal, F2) are
implemented; the Fisher / pruning / retrain baselines are registered but raise with a scope note.
"""
from .boundary_ex import boundary_expanding
from .boundary_sh import boundary_shrink
from .FT import FT, FT_l1
from .GA import GA, GA_l1
from .impl import (FusedMaskedSGD, iterative_unlearn, load_unlearn_checkpoint, save_unlearn_checkpoint)
from .RL import RL
from .RL_pro import RL_proximal


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
    "boundary_expanding": boundary_expanding, "boundary_shrink": boundary_shrink, "RL_proximal": RL_proximal,
}


def get_unlearn_method(name):
    """method usage:  function(data_loaders, model, criterion, args, mask=None)"""
    try:
        return _REGISTRY[name]
    except KeyError:
        raise NotImplementedError(f"Unlearn method {name} not implemented!") from None
