import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def max_rel_err_cols(a, b):
    """max over batch members / columns of ||a-b||_2 / ||b||_2 (norm over the N axis = -2)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    num = np.sqrt(((a - b) ** 2).sum(axis=-2))
    den = np.maximum(np.sqrt((b**2).sum(axis=-2)), 1e-300)
    return float((num / den).max())


def tridiag_block_err(t, t64, valid, back_off=0):
    """CG tridiagonals against the reference's fp64 run of the same recurrence, on the leading block up to which the
    reference's OWN fp32 run follows its fp64 run entry by entry (golden g23: `*_valid`, criterion
    |a - b| <= 1e-4 |b| + 1e-6 max |b|): max over (column, member, entry) of |t - t64| / (|t64| + 1e-2 max |t64|)
    -- 1e-4 here is the generator's criterion.  Returns (worst error, smallest block)."""
    t = np.asarray(t, dtype=np.float64)
    t64 = np.asarray(t64, dtype=np.float64)
    worst, smallest = 0.0, None
    for i in range(t64.shape[0]):
        for b in range(t64.shape[1]):
            k = min(int(valid[i, b]) - back_off, t.shape[-1], t64.shape[-1])
            assert k >= 1
            smallest = k if smallest is None else min(smallest, k)
            blk = t64[i, b, :k, :k]
            worst = max(worst, float((np.abs(t[i, b, :k, :k] - blk) / (np.abs(blk) + 1e-2 * np.abs(blk).max())).max()))
    return worst, smallest


@pytest.fixture
def legacy_resident_engines(monkeypatch):
    """Tests of the operator-resident kernels that iterate on the ROWS (lockstep, serial root / Q form, w recurrence):
    since round 5 the result-only first pass of an operator with the R-space form (lo_precond_desc.RS) runs on R + 1
    coordinates instead (csrc/lo_rspace.hip); these kernels remain the repeat with the state, the engines under the
    batch-global stop rule and the path of roots without the fp64 Gram matrices.  This switches the R-space forms off."""
    monkeypatch.setenv("LO_NO_RSPACE_COLS", "1")
    monkeypatch.setenv("LO_OC_NO_RSPACE", "1")
