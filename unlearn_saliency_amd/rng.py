"""Counter-based synthetic-input generators (splitmix64 of seed + index).

The same integer recipe exists in three places that must agree bit for bit: here
(numpy, host), csrc/salun_common.h (device; `ops.fill_*`) and oracle/salun_oracle.c
(CPU oracle).  Inputs built from it regenerate identically on any machine and torch
version, which is what lets the GPU box rebuild the benchmark's
synthetic CIFAR set and parity vectors without shipping data.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _index(n: int, seed: int, mul: int = 1, add: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        return np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(n, dtype=np.uint64) * np.uint64(mul) + np.uint64(add)


def uniform(n: int, seed: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    u = (splitmix64(_index(n, seed)) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.float32(lo) + np.float32(hi - lo) * u


def normal(n: int, seed: int, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    """Irwin-Hall(12) of 16-bit chunks: mean 0, variance 1, support (-6, 6); integer sums, so exact."""
    s = np.zeros(n, dtype=np.int64)
    for j in range(3):
        r = splitmix64(_index(n, seed, 3, j))
        for sh in (0, 16, 32, 48):
            s += ((r >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    z = (s - 393210).astype(np.float32) * np.float32(1.0 / 65536.0)
    return np.float32(mean) + np.float32(std) * z


def u8(n: int, seed: int) -> np.ndarray:
    words = splitmix64(_index((n + 7) // 8, seed))
    return words.view(np.uint8)[:n].copy()  # little-endian: byte (i & 7) of word i >> 3
