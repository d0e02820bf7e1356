"""Counter-based, platform-independent random tensors (numpy only).

Every tensor is a pure function of (name, seed, shape): element i of tensor ``name`` is
``splitmix64(fnv1a64(name) ^ seed*GOLDEN + i*GOLDEN)``.  The golden-fixture generator
(``tools/gen_golden.py``, which imports the reference), the CPU oracle tests and the GPU parity
tests all regenerate identical weights and inputs from names alone, so the committed fixtures only
hold outputs.  No torch, no global RNG state.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def fnv1a64(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _bits(name: str, n: int, seed: int, stream: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        key = fnv1a64(name) ^ (np.uint64(seed) * _GOLDEN) ^ (np.uint64(stream) * _M2)
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(key + (idx + np.uint64(1)) * _GOLDEN)


def uniform01(name: str, shape, seed: int = 0, stream: int = 0) -> np.ndarray:
    """float32 in [0, 1) with 24 random bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_bits(name, n, seed, stream) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u.reshape(shape)


def uniform(name: str, shape, lo: float, hi: float, seed: int = 0) -> np.ndarray:
    return (np.float32(lo) + np.float32(hi - lo) * uniform01(name, shape, seed)).astype(np.float32)


def normal(name: str, shape, std: float = 1.0, seed: int = 0) -> np.ndarray:
    """Box-Muller on two independent streams."""
    u1 = uniform01(name, shape, seed, stream=1).astype(np.float64)
    u2 = uniform01(name, shape, seed, stream=2).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (std * r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def randint(name: str, shape, n: int, seed: int = 0) -> np.ndarray:
    m = int(np.prod(shape)) if len(shape) else 1
    return (_bits(name, m, seed, stream=3) % np.uint64(n)).astype(np.int64).reshape(shape)
