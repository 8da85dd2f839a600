"""Flat parameter arena: the data layout every HBM-bound kernel streams over.

The reference keeps parameters, gradients, optimizer state and masks as 62 / 334 / 686
separate tensors and launches a handful of ATen kernels per tensor per step
(SURVEY.md §2.3).  Here every per-weight quantity is ONE contiguous fp32 vector (u8 for
the mask) in ``named_parameters()`` order, each tensor row-major — exactly the
concatenation the reference builds with ``torch.cat([t.flatten() ...])`` before ranking
(Classification/generate_mask.py:57), so a flat index here is the reference's flat index
and the global top-k tie rule is defined on it.

``nn.Parameter.data`` and ``.grad`` become *views* into the flat vectors, so autograd,
MIOpen and RCCL all operate on the same memory and one kernel launch (or one all-reduce)
covers the whole model.  The base allocation is 256-byte aligned by the caching
allocator; tensors are packed densely (no padding), kernels handle any tail.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.nn as nn


class FlatArena:
    """Owns flat `params` and `grads` for a module; per-tensor views keep nn.Module working."""

    def __init__(self, named_params: Iterable[Tuple[str, nn.Parameter]], device: Optional[torch.device] = None):
        named = [(n, p) for n, p in named_params]
        if not named:
            raise ValueError("no parameters")
        self.names = [n for n, _ in named]
        self._params = [p for _, p in named]
        self.device = torch.device(device) if device is not None else named[0][1].device
        self.shapes = [tuple(p.shape) for p in self._params]
        self.numels = [p.numel() for p in self._params]
        self.offsets = []
        off = 0
        for k in self.numels:
            self.offsets.append(off)
            off += k
        self.n = off
        for p in self._params:
            if p.dtype != torch.float32:
                raise TypeError("the flat arena is fp32 (the reference trains in fp32, SURVEY.md §0 fact 4)")
        self.params = torch.empty(self.n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for p, o, k, shp in zip(self._params, self.offsets, self.numels, self.shapes):
                self.params[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.params[o:o + k].view(shp)
                # weight-image caches (conv_bf16.packed_weight) key on the flat vector's version counter: an in-place
                # torch write on `arena.params` bumps that one, not the parameter's own
                p._salun_flat = self.params
        self.attach_grads()

    @classmethod
    def from_module(cls, model: nn.Module, device: Optional[torch.device] = None) -> "FlatArena":
        return cls(model.named_parameters(), device)

    # ------------------------------------------------------------------ gradients
    def attach_grads(self) -> None:
        """(Re)point every .grad at its slice of the flat gradient vector."""
        for p, o, k, shp in zip(self._params, self.offsets, self.numels, self.shapes):
            p.grad = self.grads[o:o + k].view(shp)

    def zero_grad(self) -> None:
        """One memset for the whole model (replaces optimizer.zero_grad's per-tensor loop).
        autograd then accumulates in place into the views."""
        self.grads.zero_()
        p0 = self._params[0]
        if p0.grad is None or p0.grad.data_ptr() != self.grads.data_ptr():
            self.attach_grads()

    # ---------------------------------------------------------------------- views
    def view_dict(self, flat: torch.Tensor, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
        """name -> view of `flat` with the parameter's shape (no copy)."""
        assert flat.numel() == self.n
        return OrderedDict((prefix + n, flat[o:o + k].view(shp))
                           for n, o, k, shp in zip(self.names, self.offsets, self.numels, self.shapes))

    def pack(self, tensors: Dict[str, torch.Tensor], prefix: str = "") -> torch.Tensor:
        """name -> fp32 tensor dict (e.g. the Fisher dictionary of DDPM/runners/diffusion.py:177-191) -> one flat
        device vector in arena order.  Every parameter must be present with its shape."""
        out = torch.empty(self.n, dtype=torch.float32, device=self.device)
        for n, o, k, shp in zip(self.names, self.offsets, self.numels, self.shapes):
            t = tensors[prefix + n]
            if tuple(t.shape) != tuple(shp):
                raise ValueError(f"{prefix + n}: shape {tuple(t.shape)} != parameter shape {tuple(shp)}")
            out[o:o + k] = t.reshape(-1).to(device=self.device, dtype=torch.float32)
        return out

    def new_like(self, dtype=torch.float32, zero: bool = True) -> torch.Tensor:
        f = torch.zeros if zero else torch.empty
        return f(self.n, dtype=dtype, device=self.device)

    # ----------------------------------------------------------------------- masks
    def pack_mask(self, mask: Dict[str, torch.Tensor], prefix: str = "", strict: bool = True) -> torch.Tensor:
        """Reference mask dict (name -> int64/any 0/1 tensor of the parameter's shape, on any
        device; Classification/generate_mask.py:76-82) -> flat u8 device vector.
        Done once per run instead of the reference's per-step `mask[name].to(device)`
        (DDPM/runners/diffusion.py:589-592)."""
        from . import ops
        out = torch.empty(self.n, dtype=torch.uint8, device=self.device)
        for n, o, k, shp in zip(self.names, self.offsets, self.numels, self.shapes):
            key = prefix + n
            if key not in mask:
                if strict:
                    raise KeyError(f"mask has no entry for parameter {key!r}")
                out[o:o + k] = 1
                continue
            t = mask[key]
            if tuple(t.shape) != shp:
                raise ValueError(f"mask[{key!r}] has shape {tuple(t.shape)}, parameter has {shp}")
            t = t.to(self.device)
            if t.dtype == torch.int64 and t.is_contiguous():
                ops.mask_i64_to_u8(t.reshape(-1), out[o:o + k])
            else:
                out[o:o + k] = (t.reshape(-1) != 0).to(torch.uint8)
        return out

    def unpack_mask(self, flat_u8: torch.Tensor, prefix: str = "", device: Optional[torch.device] = None
                    ) -> "OrderedDict[str, torch.Tensor]":
        """flat u8 -> the reference's artefact: dict of int64 0/1 tensors with parameter shapes."""
        from . import ops
        i64 = ops.mask_u8_to_i64(flat_u8)
        if device is not None:
            i64 = i64.to(device)
        return OrderedDict((prefix + n, i64[o:o + k].view(shp).clone())
                           for n, o, k, shp in zip(self.names, self.offsets, self.numels, self.shapes))


_ARENA_ATTR = "_salun_flat_arena"


def arena_of(model: nn.Module, device: Optional[torch.device] = None) -> FlatArena:
    """The model's arena, created on first use and cached on the module.  Re-created if the
    parameters were re-homed behind our back (e.g. `model.to(...)`, load_state_dict is fine)."""
    a = getattr(model, _ARENA_ATTR, None)
    params = list(model.parameters())
    if a is not None and len(params) == len(a._params) and all(
            p is q and p.data_ptr() == a.params.data_ptr() + 4 * o
            for p, q, o in zip(params, a._params, a.offsets)):
        return a
    a = FlatArena.from_module(model, device)
    object.__setattr__(model, _ARENA_ATTR, a)
    return a
