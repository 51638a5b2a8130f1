"""Seeded synthetic inputs shared by the golden-vector generator (make_golden.py, which runs the
real reference in the build container) and by the parity tests (which run the oracle / the HIP
path on the very same inputs).  numpy PCG64 streams only -- no torch RNG -- so the inputs can be
regenerated anywhere; every golden file also stores a float64 checksum of its inputs so that a
drifting RNG would be detected instead of silently comparing different problems.
"""
from __future__ import annotations

import numpy as np


def _rng(seed):
    return np.random.default_rng(seed)


def checksum(*arrays):
    return float(sum(np.asarray(a, dtype=np.float64).sum() + 3.0 * np.abs(np.asarray(a, dtype=np.float64)).sum()
                     for a in arrays))


# ---- recipes from the reference's own unit tests -------------------------------------------------

def spd_test_matrix(seed, size, batch=(), dtype=np.float64, jitter=1e-1):
    """test/utils/test_linear_cg.py:27-32 recipe: X X^T / ||.||_F + 0.1 I  (norm over the whole tensor)."""
    g = _rng(seed)
    X = g.standard_normal(tuple(batch) + (size, size))
    M = X @ np.swapaxes(X, -1, -2)
    M = M / np.sqrt((M * M).sum())
    M = M + jitter * np.eye(size)
    return M.astype(dtype)


def randn(seed, *shape, dtype=np.float64):
    return _rng(seed).standard_normal(shape).astype(dtype)


# ---- benchmark-shaped synthetic operators ------------------------------

def lowrank_diag(seed, B, N, R, c, dtype=np.float32):
    """C = randn/sqrt(R), d = rand + 0.5, rhs = randn  (cfg2/cfg3 recipe)."""
    g = _rng(seed)
    bs = tuple(B) if isinstance(B, (tuple, list)) else (B,)
    C = (g.standard_normal(bs + (N, R)) / np.sqrt(R)).astype(dtype)
    d = (g.random(bs + (N,)) + 0.5).astype(dtype)
    rhs = g.standard_normal(bs + (N, c)).astype(dtype)
    return C, d, rhs


def probes(seed, B, N, P, dtype=np.float32):
    """Unit-norm probe columns + their norms, the pair `_probe_vectors_and_norms` returns."""
    g = _rng(seed)
    bs = tuple(B) if isinstance(B, (tuple, list)) else (B,)
    Z = g.standard_normal(bs + (N, P)).astype(dtype)
    nrm = np.sqrt((Z * Z).sum(axis=-2, keepdims=True, dtype=dtype))
    return (Z / nrm).astype(dtype), nrm.astype(dtype)


def kron_factors(seed, B, n1, n2, c, sigma=1e-2, dtype=np.float32):
    """K_i = X X^T + 0.1 I with X = randn / sqrt(n_i)  (cfg4 recipe scaled to the factor size)."""
    g = _rng(seed)
    bs = (B,)
    X1 = g.standard_normal(bs + (n1, n1)) / np.sqrt(n1)
    X2 = g.standard_normal(bs + (n2, n2)) / np.sqrt(n2)
    K1 = (X1 @ np.swapaxes(X1, -1, -2) + 0.1 * np.eye(n1)).astype(dtype)
    K2 = (X2 @ np.swapaxes(X2, -1, -2) + 0.1 * np.eye(n2)).astype(dtype)
    sig = np.full(bs + (1,), sigma, dtype=dtype)
    rhs = g.standard_normal(bs + (n1 * n2, c)).astype(dtype)
    return K1, K2, sig, rhs


def dense_diag(seed, B, N, c, dtype=np.float32):
    """K = X X^T with X = randn / sqrt(N), d = rand + 0.5  (cfg5 recipe)."""
    g = _rng(seed)
    bs = (B,)
    X = (g.standard_normal(bs + (N, N)) / np.sqrt(N)).astype(dtype)
    K = (X @ np.swapaxes(X, -1, -2)).astype(dtype)
    K = ((K + np.swapaxes(K, -1, -2)) * dtype(0.5)).astype(dtype)
    d = (g.random(bs + (N,)) + 0.5).astype(dtype)
    rhs = g.standard_normal(bs + (N, c)).astype(dtype)
    return K, d, rhs


def pivchol_dense8(seed, batch=()):
    """test/functions/test_pivoted_cholesky.py:24-27 recipe: mat = randn(8,8); mat @ mat.mT."""
    X = randn(seed, *batch, 8, 8, dtype=np.float32)
    return (X @ np.swapaxes(X, -1, -2)).astype(np.float32)
