"""Deterministic, platform-independent pseudo-random tensors (splitmix64 on the flat index).

Used by make_golden.py (generator) and by the tests (consumer) so that large inputs need not be stored
in the fixtures: only the seed and the expected outputs are.  Pure integer arithmetic in uint64 followed
by one exact conversion, so the values are bit-identical on every machine.
"""
import numpy as np


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array, uniform in [lo, hi) on a 2^-24 grid (exactly representable)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        bits = _splitmix64(idx) >> np.uint64(40)           # 24 random bits
    u = bits.astype(np.float64) / float(1 << 24)           # [0,1) exact
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def det_normalish(shape, seed, std=1.0):
    """Sum of 4 uniforms, rescaled: bell-shaped, bounded, deterministic."""
    a = sum(det_uniform(shape, seed * 7 + k, -1.0, 1.0).astype(np.float64) for k in range(4))
    return (a * (std * (3.0 / 4.0) ** 0.5)).astype(np.float32)


def det_int(shape, seed, n):
    n_el = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n_el, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        bits = _splitmix64(idx) >> np.uint64(33)
    return (bits % np.uint64(n)).astype(np.int64).reshape(shape)
