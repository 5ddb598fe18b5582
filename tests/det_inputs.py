"""Deterministic test inputs by integer formula (no RNG state, no torch generator).

The reference-generated fixtures (tests/golden/ref_*.npz, made by tests/golden/make_ref_fixtures.py)
store the small inputs and the reference's outputs; the large inputs (paged KV caches, packed
weight matrices) are rebuilt from these formulas on both sides and verified by the CRC32 the
fixture stores, so a fixture stays a few hundred KB while the data is bit-identical everywhere.
Pure numpy uint64 arithmetic (SplitMix64 finaliser) + exactly rounded float32 scaling.
"""
import zlib

import numpy as np
import torch

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def u32(n: int, seed: int) -> np.ndarray:
    """n pseudo-random 32-bit words, a pure function of (index, seed)."""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)) * _M1
        x ^= x >> np.uint64(30)
        x *= _M2
        x ^= x >> np.uint64(27)
        x *= _M3
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(32)).astype(np.uint32)


def uniform(shape, seed: int, lo: float, hi: float) -> torch.Tensor:
    """float32 tensor, U[lo, hi) on a 2^-24 grid (every step exactly rounded in float32)."""
    n = int(np.prod(shape))
    u = (u32(n, seed) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    v = u * np.float32(hi - lo) + np.float32(lo)
    return torch.from_numpy(v.reshape(shape).copy())


def normalish(shape, seed: int, std: float = 1.0) -> torch.Tensor:
    """float32, sum of four uniforms (Irwin-Hall, variance-normalised): bell-shaped, bounded by
    +-3.46 std — stands in for randn without any library RNG."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float32)
    for j in range(4):
        acc += (u32(n, seed * 4 + j) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    v = (acc - np.float32(2.0)) * np.float32(std * (3.0 ** 0.5))
    return torch.from_numpy(v.reshape(shape).copy())


def int32_words(shape, seed: int) -> torch.Tensor:
    """Random int32 words (all 32 bits used) — packed 4-bit weights / zeros."""
    n = int(np.prod(shape))
    return torch.from_numpy(u32(n, seed).view(np.int32).reshape(shape).copy())


def randint(shape, seed: int, lo: int, hi: int) -> torch.Tensor:
    """int64 in [lo, hi)."""
    n = int(np.prod(shape))
    v = (u32(n, seed).astype(np.int64) % (hi - lo)) + lo
    return torch.from_numpy(v.reshape(shape).copy())


def crc(t: torch.Tensor) -> int:
    t = t.detach().contiguous()
    view = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()]
    return zlib.crc32(t.view(view).numpy().tobytes())
