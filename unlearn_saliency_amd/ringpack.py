"""Packed weight images for the LDS-DMA ring convolution (csrc/salun_conv_ring.hip, K8r).

`conv.use_salun_convs(model)` registers the weight of every 3x3 / stride 1 / pad 1 convolution here; `images(w)` then
hands `ops.conv2d_forward` / `ops.conv2d_backward_data` the forward and backward-data images of that weight, re-packed —
ALL stale registered weights of the group in ONE launch — when the parameters changed since the last pack:
`ops.PARAM_EPOCH` (raw-pointer writes of the fused optimizer kernels), torch's version counter of the parameter and of
the flat arena it is a view of (flat.py).  Only registered parameters are served: a cache keyed by address alone would
hand a freed-and-reused address a stale image.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Optional

import torch

from . import _lib


class _Entry:
    __slots__ = ("ref", "ptr", "shape", "key", "img_f", "img_d", "group", "flat")

    def __init__(self, p: torch.Tensor, group: "Group"):
        self.ref = weakref.ref(p)
        self.ptr, self.shape = p.data_ptr(), tuple(p.shape)
        self.key = None
        self.img_f = self.img_d = None
        self.group = group
        self.flat = getattr(p, "_salun_flat", None)


class Group:
    """The registered weights of one model: packed together."""

    def __init__(self):
        self.entries: list[_Entry] = []
        self.event: Optional[torch.cuda.Event] = None
        self.stream: int = 0


_all: list[_Entry] = []            # every registered weight
_by_ptr: dict[int, _Entry] = {}    # current address -> entry (rebuilt when a lookup misses: FlatArena re-homes parameters)
_not_ours: set[int] = set()        # addresses looked up and found unregistered since the last rebuild
ENABLED = [True]       # tools / tests: A/B switch
PACK_LAUNCHES = [0]    # launches of the pack kernel so far (tests, host profile)


def eligible(K: int, C: int, R: int, stride: int, pad: int) -> bool:
    return R == 3 and stride == 1 and pad == 1 and C % 8 == 0 and K % 8 == 0


def register(params) -> Optional[Group]:
    """Register the OIHW [K, C, 3, 3] weights `params` (nn.Parameters that outlive their use) as one pack group."""
    g = Group()
    known = {id(e.ref()) for e in _all if e.ref() is not None}
    for p in params:
        if not (p.dtype == torch.float32 and p.dim() == 4):   # (the device is looked at when an image is asked for)
            continue
        K, C, R, _ = p.shape
        if not eligible(K, C, R, 1, 1) or p.shape[3] != 3 or id(p) in known:
            continue
        e = _Entry(p, g)
        g.entries.append(e)
        _all.append(e)
        known.add(id(p))
    _rebuild()
    return g if g.entries else None


def _rebuild() -> None:
    _by_ptr.clear()
    _not_ours.clear()
    live = []
    for e in _all:
        p = e.ref()
        if p is None:
            continue
        if p.data_ptr() != e.ptr:       # re-homed (a flat arena was built after registration): images are stale
            e.ptr, e.key = p.data_ptr(), None
        e.flat = getattr(p, "_salun_flat", None)
        _by_ptr[e.ptr] = e
        live.append(e)
    _all[:] = live


def _lookup(w: torch.Tensor) -> Optional[_Entry]:
    ptr = w.data_ptr()
    e = _by_ptr.get(ptr)
    if e is not None:
        p = e.ref()
        if p is not None and p.data_ptr() == ptr and tuple(w.shape) == e.shape:
            return e
        _rebuild()                      # the parameter died or moved: never serve its old address again
        e = _by_ptr.get(ptr)
        return e if e is not None and tuple(w.shape) == e.shape else None
    if ptr in _not_ours:
        return None
    _rebuild()
    e = _by_ptr.get(ptr)
    if e is None or tuple(w.shape) != e.shape:
        _not_ours.add(ptr)
        return None
    return e


_ops_mod = []


def _ops():
    if not _ops_mod:
        from . import ops   # (ops imports this module)
        _ops_mod.append(ops)
    return _ops_mod[0]


def _key(e: _Entry, p: torch.Tensor, epoch: int):
    return (epoch, p._version, e.flat._version if e.flat is not None else -1)


def _repack(g: Group, epoch: int) -> None:
    L = _lib.lib()
    jobs = []
    for e in g.entries:
        p = e.ref()
        if p is None or not p.is_cuda:
            continue
        if p.data_ptr() != e.ptr:       # moved since the last lookup of THIS weight: re-key, its image is stale
            _by_ptr.pop(e.ptr, None)
            e.ptr, e.key = p.data_ptr(), None
            e.flat = getattr(p, "_salun_flat", None)
            _by_ptr[e.ptr] = e
        k = _key(e, p, epoch)
        if e.key == k and e.img_f is not None:
            continue
        K, C = e.shape[0], e.shape[1]
        if e.img_f is None or e.img_f.device != p.device:
            e.img_f = torch.empty(int(L.salun_conv3x3_pack_bytes(K, C, 0)) // 4, dtype=torch.float32, device=p.device)
            e.img_d = torch.empty(int(L.salun_conv3x3_pack_bytes(K, C, 1)) // 4, dtype=torch.float32, device=p.device)
        e.key = k
        jobs.append(_lib.PackJob(p.data_ptr(), e.img_f.data_ptr(), e.img_d.data_ptr(), K, C))
    if not jobs:
        return
    ops = _ops()
    arr = (_lib.PackJob * len(jobs))(*jobs)
    _lib.check(L.salun_conv3x3_pack_weights(ctypes.cast(arr, ctypes.c_void_p), len(jobs), ops._stream()),
               "salun_conv3x3_pack_weights")
    PACK_LAUNCHES[0] += (len(jobs) + 31) // 32
    g.stream = ops._stream_handle()
    if g.event is None:
        g.event = torch.cuda.Event()
    g.event.record()


def images(w: torch.Tensor):
    """(forward image, backward-data image) of a registered weight, current with the parameters; None if `w` is not
    registered (the caller then runs conv_igemm on the OIHW tensor itself)."""
    if not ENABLED[0]:
        return None
    e = _lookup(w)
    if e is None:
        return None
    p = e.ref()
    if not w.is_cuda:
        return None
    ops = _ops()
    epoch = ops.PARAM_EPOCH[0]
    g = e.group
    if e.key != _key(e, p, epoch):
        _repack(g, epoch)
    elif g.event is not None and g.stream != ops._stream_handle():
        # packed on another stream (the no-grad target pass of the diffusion steps runs beside the forget pass)
        torch.cuda.current_stream().wait_event(g.event)
    return e.img_f, e.img_d
