"""Per-batch randomness under data parallel: every draw is a function of the GLOBAL batch.

The reference is one process feeding `nn.DataParallel` (DDPM/runners/diffusion.py:504,948): noise, timesteps and the
label-drop mask are drawn once for the whole batch (:530-533, models/diffusion.py:340-343) and every replica's
`nn.Dropout` draws on its own chunk.  Here a batch of b samples is split over ranks as contiguous shards [lo, hi); all
ranks hold identically seeded generators and

  * generator draws (noise, timesteps, label drop) are taken for the GLOBAL batch on every rank and sliced to the
    shard — ranks consume the generators in lock-step whatever their shard sizes;
  * dropout keep-decisions come from a counter-based generator keyed by (step, call index, global sample index,
    position) — `salun_dropout` — so a rank computes only its own rows and the concatenation over ranks IS the
    single-process mask.

`scope(lo, hi, b)` tells the model code which shard the tensors of the current forward pass belong to; with no scope
(or b == local batch) everything reduces to plain single-process draws in the same generator order as before.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import dist as sdist

_M64 = (1 << 64) - 1


class Shard:
    """Samples [lo, hi) of a global batch of b.  `sliced` is False for a batch that belongs to this rank alone (single
    process, or loaders that give every rank its own batches): plain draws, loss weight 1."""
    __slots__ = ("lo", "hi", "b", "sliced")

    def __init__(self, lo: int, hi: int, b: int, sliced: bool = True):
        self.lo, self.hi, self.b, self.sliced = int(lo), int(hi), int(b), bool(sliced)

    @property
    def n(self) -> int:
        return self.hi - self.lo

    @property
    def weight(self) -> float:
        """shard-mean loss -> share of the global-batch mean under the AVG all-reduce of the gradients"""
        ws = sdist.world_size()
        return 1.0 if (ws <= 1 or not self.sliced) else self.n * ws / float(self.b)

    @property
    def share(self) -> float:
        """shard-mean loss -> share of the global-batch mean when the gradients are SUMmed over ranks (Phase A)"""
        return 1.0 if not self.sliced else self.n / float(self.b)


_current: Optional[Shard] = None


def current() -> Optional[Shard]:
    return _current


class scope:
    """`with draws.scope(shard):` — model calls inside belong to that shard of a global batch."""

    def __init__(self, shard: Optional[Shard]):
        self.shard = shard

    def __enter__(self):
        global _current
        self._prev = _current
        _current = self.shard
        return self.shard

    def __exit__(self, *exc):
        global _current
        _current = self._prev
        return False


def shard_of(loader, n_local: int) -> Shard:
    """The shard the loader just yielded (`loader.last_shard`), or the trivial one."""
    sh = getattr(loader, "last_shard", None) if sdist.world_size() > 1 else None
    lo, hi, b = sh if sh is not None else (0, n_local, n_local)
    assert hi - lo == n_local, (sh, n_local)
    return Shard(lo, hi, b, sliced=sh is not None)


def _sliced(shard: Optional[Shard], n_local: int) -> Optional[Shard]:
    if shard is None or not shard.sliced or shard.b == n_local:
        return None
    assert shard.n == n_local, (shard.lo, shard.hi, shard.b, n_local)
    return shard


def randn_like(x: torch.Tensor, shard: Optional[Shard] = None) -> torch.Tensor:
    sh = _sliced(shard if shard is not None else _current, x.size(0))
    if sh is None:
        return torch.randn_like(x)
    return torch.randn((sh.b,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)[sh.lo:sh.hi]


def randint(high: int, n_local: int, device, shard: Optional[Shard] = None) -> torch.Tensor:
    """`torch.randint(0, high, (n,), device=device)` for the global batch, sliced."""
    sh = _sliced(shard if shard is not None else _current, n_local)
    if sh is None:
        return torch.randint(0, high, (n_local,), device=device)
    return torch.randint(0, high, (sh.b,), device=device)[sh.lo:sh.hi]


def batch_draw(n_local: int, fn, shard: Optional[Shard] = None) -> torch.Tensor:
    """`fn(n)` draws one value per sample (first dimension n): evaluated for the global batch and sliced."""
    sh = _sliced(shard if shard is not None else _current, n_local)
    if sh is None:
        return fn(n_local)
    return fn(sh.b)[sh.lo:sh.hi]


# ----------------------------------------------------------------------------- dropout keys
# key(step, call) — identical on every rank (same base seed, same step count, same host call order within a step).  The
# step counter is advanced by the training loops (`next_step`), so a rank that skipped a pass (empty shard of a ragged
# tail batch) is back in lock-step at the next step.
_base: Optional[int] = None
_step = 0
_call = 0


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def seed(value: Optional[int]) -> None:
    """Base seed of the dropout keys (None: follow `torch.initial_seed()`, i.e. `torch.manual_seed`)."""
    global _base, _step, _call
    _base = None if value is None else int(value) & _M64
    _step = _call = 0


def next_step() -> int:
    global _step, _call
    _step += 1
    _call = 0
    return _step


def dropout_key() -> Tuple[int, int]:
    """-> (key, sample_offset) for the next dropout call of the current forward pass."""
    global _call
    base = _base if _base is not None else (torch.initial_seed() & _M64)
    k = _splitmix64((_splitmix64(base) + _step) & _M64)
    key = (k + (_call << 40)) & _M64
    _call += 1
    if _current is not None and _current.sliced:
        return key, _current.lo
    if sdist.world_size() > 1:  # every rank has its own batch: independent masks per rank
        key = _splitmix64((key + sdist.rank()) & _M64)
    return key, 0


def state() -> Tuple[Optional[int], int, int]:
    return _base, _step, _call


def set_state(st) -> None:
    global _base, _step, _call
    _base, _step, _call = st


# ----------------------------------------------------------------------------- modules
class CounterDropout(torch.nn.Dropout):
    """`nn.Dropout` of the diffusion ResnetBlock (reference DDPM/models/diffusion.py:97,124) on the counter-based
    generator: fp32 device activations in training mode go through `salun_dropout` with this forward pass's key and the
    shard's global sample offset.  Host tensors (the model code exercised on the CPU by tests) use torch's generator,
    drawn for the global batch and sliced like every other draw of this module."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training or self.p == 0.0:
            return x
        if x.is_cuda and x.dtype == torch.float32 and not self.inplace:
            from . import ops
            key, off = dropout_key()
            return ops.dropout_fn(x, self.p, key, off)
        sh = _sliced(_current, x.size(0))
        lo, hi, b = (sh.lo, sh.hi, sh.b) if sh is not None else (0, x.size(0), x.size(0))
        keep = torch.rand((b,) + tuple(x.shape[1:]), device=x.device) >= self.p
        return x * keep[lo:hi].to(x.dtype) / (1.0 - self.p)
