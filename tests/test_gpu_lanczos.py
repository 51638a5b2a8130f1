"""Explicit Lanczos (reference: utils/lanczos.py:9-164): goldens at 1e-4 on the leading block where the reference's fp32 and
fp64 runs agree (g27), the basis as a view of the step kernels' layout and the native-layout root epilogue."""
import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, max_rel_err_cols

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from oracle import lo_oracle as orc  # noqa: E402  (the checker)
from oracle import lo_oracle_c as occ  # noqa: E402  (the checker, C restatement)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------ Lanczos at 1e-4 where the reference's fp32 and fp64 runs agree (g27)
def _leading_agreement(a32, a64, tol):
    """Largest k such that the reference's own float32 run is within tol of its float64 run on every leading index < k
    (last axis), measured per index against the scale of the float64 values."""
    a32, a64 = np.asarray(a32, dtype=np.float64), np.asarray(a64, dtype=np.float64)
    k = 0
    for j in range(min(a32.shape[-1], a64.shape[-1])):
        if np.abs(a32[..., : j + 1] - a64[..., : j + 1]).max() > tol:
            break
        k = j + 1
    return k

@pytest.mark.parametrize("case", ["batch", "cfg3"])
def test_lanczos_at_1e4_on_the_block_where_the_reference_agrees_with_itself(case):
    g = load_golden("g27_lanczos_divergence")
    if case == "batch":
        C, d, _ = cases.lowrank_diag(511, 2, 256, 8, 1)
        V, steps = cases.randn(512, 2, 256, 3, dtype=np.float32), 10
    else:
        C, d, _ = cases.lowrank_diag(2701, 2, 2048, 32, 1)
        V, steps = cases.randn(2702, 2, 2048, 4, dtype=np.float32), 20
    q, t = K.lanczos_tridiag(K.lowrank_diag_descriptor(dev(C), dev(d)), dev(V), steps)
    q, t = host(q).astype(np.float64), host(t).astype(np.float64)
    t32, t64, q32, q64 = g[f"t_{case}_f32"], g[f"t_{case}_f64"], g[f"q_{case}_f32"], g[f"q_{case}_f64"]
    assert q.shape == q64.shape and t.shape == t64.shape
    scale = np.abs(t64).max()
    # tridiagonals: entries (i, j <= k) -- the reference's float32 run within 1e-5 of its float64 run there
    colerr = np.abs(np.asarray(t32, np.float64) - t64).max(-2) / scale  # worst entry of every column
    kt = _leading_agreement(colerr, np.zeros_like(colerr), 1e-5)
    assert kt >= steps // 2, kt
    assert np.abs(t[..., :kt, :kt] - t64[..., :kt, :kt]).max() <= 1e-4 * scale
    # basis vectors (unit columns): column error in the 2-norm
    e32 = np.sqrt(((np.asarray(q32, np.float64) - q64) ** 2).sum(-2))
    kq = _leading_agreement(e32, np.zeros_like(e32), 3e-5)
    assert kq >= steps // 2, kq
    eh = np.sqrt(((q - q64) ** 2).sum(-2))
    assert eh[..., :kq].max() <= 1e-4, eh[..., :kq].max()


import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, max_rel_err_cols, tridiag_block_err

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from linear_operator_amd import settings  # noqa: E402
from linear_operator_amd.operators import (  # noqa: E402
    AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator, DiagLinearOperator,
    KroneckerProductLinearOperator, LowRankRootLinearOperator,
)
from oracle import lo_oracle as orc  # noqa: E402  (the checker)


class ProbedAddedDiag(AddedDiagLinearOperator):
    _probes = None

    def _probe_vectors_and_norms(self):  # hook: reference _linear_operator.py:629-633
        return self._probes


# ---------------------------------------------------------------- Lanczos basis without the layout copy
@pytest.mark.parametrize("B,N,P,k", [(6, 3000, 16, 20), (3, 1000, 4, 12), (2, 700, 1, 9), (5, 2048, 8, 32)])
def test_lanczos_basis_view_and_native_root_epilogue(B, N, P, k):
    """`lanczos_tridiag` returns q_mat [P, B, N, k] as a VIEW of the basis in the step kernels' layout [k, B, N, P]
    (no 5 GB copy at the cfg3 shape); `root_from_lanczos` reads that layout directly.  Same values as the reference
    layout (lo_lanczos_permute_f32), bit-identical epilogue outputs."""
    C, d, _ = cases.lowrank_diag(9900 + P, B, N, 16, 1)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    V = dev(cases.randn(9901, B, N, P, dtype=np.float32))
    q_view, t_view = K.lanczos_tridiag(desc, V, k)
    q_cont, t_cont = K.lanczos_tridiag(desc, V, k, contiguous=True)
    assert q_view.shape == q_cont.shape and not q_view.is_contiguous() and q_cont.is_contiguous()
    assert torch.equal(q_view, q_cont) and torch.equal(t_view, t_cont)
    if P == 1:
        q_view, q_cont, t_view = q_view.unsqueeze(0), q_cont.unsqueeze(0), t_view.unsqueeze(0)
    assert (K._native_lanczos_layout(q_view) is not None) and K._native_lanczos_layout(q_cont) is None
    from linear_operator_amd.utils.lanczos import lanczos_tridiag_to_diag
    evals, evecs = lanczos_tridiag_to_diag(t_view + 1e-3 * torch.eye(t_view.shape[-1], device="cuda"))
    K._hip.prof_enable(True)
    a = K.root_from_lanczos(q_view, evecs, evals, want_root=True, want_inverse=True)
    torch.cuda.synchronize()
    prof = K._hip.prof_report()
    K._hip.prof_enable(False)
    assert list(prof) == ["lz_root"], sorted(prof)  # (one launch, no copy kernel in front of it)
    b = K.root_from_lanczos(q_cont, evecs, evals, want_root=True, want_inverse=True)
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y)
    ref = (q_cont.double() @ evecs.double())
    assert float((a[0].double() - ref).abs().max()) < 1e-5
