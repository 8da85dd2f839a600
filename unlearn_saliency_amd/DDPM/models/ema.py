"""Exponential moving average of the parameters — same surface as the reference's `EMAHelper`
(DDPM/models/ema.py:5-51: register / update / ema / ema_copy / state_dict / load_state_dict, `shadow` a
name -> tensor dict, DataParallel wrappers unwrapped).

The shadow is ONE flat fp32 vector in `named_parameters()` order (the arena's layout): `update` is a single
`lerp_` over it when the module's parameters live in a flat arena (one launch for 38.6 M weights instead of 334 x 3),
and a per-tensor loop over views of it otherwise.  `state_dict()` hands out per-name views, so checkpoints keep the
reference's format.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn


def _unwrap(module):
    return module.module if isinstance(module, nn.DataParallel) else module


class EMAHelper(object):
    def __init__(self, mu=0.999):
        self.mu = mu
        self.shadow = {}
        self._flat = None
        self._names = []

    def _trainable(self, module):
        return [(n, p) for n, p in _unwrap(module).named_parameters() if p.requires_grad]

    def register(self, module):
        named = self._trainable(module)
        total = sum(p.numel() for _, p in named)
        dev = named[0][1].device if named else torch.device("cpu")
        self._flat = torch.empty(total, dtype=torch.float32, device=dev)
        self._names = [n for n, _ in named]
        self.shadow, off = OrderedDict(), 0
        for n, p in named:
            k = p.numel()
            view = self._flat[off:off + k].view(p.shape)
            view.copy_(p.data)
            self.shadow[n] = view
            off += k

    def _arena_params(self, module):
        """The module's flat parameter vector when it lines up with the shadow (all parameters trainable, arena order)."""
        a = getattr(_unwrap(module), "_salun_flat_arena", None)
        if (a is not None and self._flat is not None and a.n == self._flat.numel() and a.names == self._names
                and a.params.device == self._flat.device):
            return a.params
        return None

    def update(self, module):
        flat = self._arena_params(module)
        if flat is not None:
            self._flat.lerp_(flat, 1.0 - self.mu)  # shadow = mu * shadow + (1 - mu) * param, one pass
            return
        for name, param in self._trainable(module):
            self.shadow[name].mul_(self.mu).add_(param.data, alpha=1.0 - self.mu)

    def ema(self, module):
        for name, param in self._trainable(module):
            param.data.copy_(self.shadow[name])

    def ema_copy(self, module):
        inner = _unwrap(module)
        clone = type(inner)(inner.config).to(next(inner.parameters()).device)
        clone.load_state_dict(inner.state_dict())
        self.ema(clone)
        return nn.DataParallel(clone) if isinstance(module, nn.DataParallel) else clone

    def state_dict(self):
        return self.shadow

    def load_state_dict(self, state_dict):
        if self._flat is not None and list(state_dict.keys()) == self._names:
            for n, v in state_dict.items():
                self.shadow[n].copy_(v)
        else:
            self.shadow = state_dict
            self._flat, self._names = None, list(state_dict.keys())
