"""Shared synthetic-input builders for the parity tests (seeded, small)."""
import numpy as np


def rand_codes(rng, n, M):
    return rng.integers(0, 256, (n, M // 2), dtype=np.uint8)


def rand_qtables(rng, shape_prefix, M, tmax):
    return rng.integers(0, tmax + 1, tuple(shape_prefix) + (M, 16)).astype(np.int8)


def float_tables(rng, nq, ma, M, scale=1.0, negatives=False):
    """Distance-table-like floats: squared distances of N(0,1) sub-vectors (sq_dim 8)."""
    d = 8
    q = rng.normal(size=(nq, ma, M, 1, d)).astype(np.float32)
    c = rng.normal(size=(1, 1, M, 16, d)).astype(np.float32)
    t = ((q - c) ** 2).sum(-1).astype(np.float32) * np.float32(scale)
    if negatives:
        t = t - np.float32(0.05) * rng.random(t.shape).astype(np.float32) * (rng.random(t.shape) < 0.02)
    return np.ascontiguousarray(t.reshape(nq, ma, M * 16), np.float32)


def heaps_equal(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wraps mod 2^64)."""
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def path_independent(fn):
    """Marks a GPU test that involves no query path (the stateless database-build entry points): tests/conftest.py then runs it
    once instead of once per scan path."""
    fn._path_independent = True
    return fn
