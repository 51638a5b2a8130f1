"""GPU tests of the two round-3 kernels of the Kronecker path (BASELINE cfg4):

* k_precond_fused_kron (csrc/lo_precond_fused.hip) -- the CG step with the preconditioner in KRONECKER ROOT FORM: the rows
  of the Woodbury factor are formed on the fly from the pivot rows of the two factors (lo_precond_kron_root_f32) instead of
  streaming Q.  Same preconditioner (added_diag_linear_operator.py:135-184 on the pivoted Cholesky of
  kronecker_product_linear_operator.py:34-45), so: same iteration counts, solutions to fp32 rounding, the reference's golden
  vectors within the north_star bar.
* k_kron_fused (csrc/lo_kron.hip) -- both GEMMs of the Kronecker matvec in one launch, the intermediate in the accumulators.
"""
import os
from unittest import mock

import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, max_rel_err_cols

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from linear_operator_amd import settings  # noqa: E402
from linear_operator_amd.operators import (  # noqa: E402
    AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator, KroneckerProductLinearOperator,
)
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo  # noqa: E402

TOL = 1e-4  # north_star bar on fp32 solves, relative per column


def _spd_factors(seed, B, n1, n2, kind="random"):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    if kind == "random":
        X1 = torch.randn(B, n1, n1, generator=g, device="cuda") / n1 ** 0.5
        X2 = torch.randn(B, n2, n2, generator=g, device="cuda") / n2 ** 0.5
        K1 = X1 @ X1.mT + 0.1 * torch.eye(n1, device="cuda")
        K2 = X2 @ X2.mT + 0.1 * torch.eye(n2, device="cuda")
    else:  # RBF kernels on 1-d inputs: smooth, the pivot rows are close to dependent
        t1 = torch.rand(B, n1, 1, generator=g, device="cuda")
        t2 = torch.rand(B, n2, 1, generator=g, device="cuda")
        K1 = torch.exp(-0.5 * (t1 - t1.mT) ** 2 / 0.6 ** 2)
        K2 = torch.exp(-0.5 * (t2 - t2.mT) ** 2 / 0.6 ** 2)
    rhs = torch.randn(B, n1 * n2, 1, generator=g, device="cuda")
    return K1, K2, rhs


def _prof_names(fn):
    K._hip.prof_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        names = set(K._hip.prof_report())
    finally:
        K._hip.prof_enable(False)
    return out, names


@pytest.mark.parametrize("B,n1,n2,expect_root", [
    (6, 128, 128, True),    # 4 rows per thread, groups of 16 workgroups
    (40, 256, 256, True),   # 8 rows per thread, groups of 32
    (3, 256, 64, True),     # n2 = 64: two first-factor indices per 128-row block
    (5, 160, 128, True),    # n1 not a power of two, N = 20480 not a multiple of the workgroup rows
    (4, 96, 96, False),     # n2 does not divide 256: the Q form stays
])
def test_kron_root_cg_matches_the_q_form(B, n1, n2, expect_root):
    K1, K2, rhs = _spd_factors(100 + n1 + n2, B, n1, n2)
    sig = torch.full((B,), 1e-2, device="cuda")
    desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
    L, perm = K.pivoted_cholesky(desc.without_diag(), 15, contiguous=False)
    pre_q = K.precond_build(L, sig, True)
    pre_k = K.precond_build(L, sig, True, perm=perm, kron=desc)
    assert pre_q.kron is None
    assert pre_k.kron is not None and pre_k.kron_kappa < 50.0
    ref, names_q = _prof_names(lambda: K.cg_solve(desc, rhs, precond=pre_q, tolerance=1e-3))
    out, names_k = _prof_names(lambda: K.cg_solve(desc, rhs, precond=pre_k, tolerance=1e-3))
    assert "precond_fused" in names_q and "precond_fused_kron" not in names_q
    assert ("precond_fused_kron" in names_k) == expect_root
    assert abs(out.iterations - ref.iterations) <= 1
    assert out.tolerance_reached and ref.tolerance_reached
    # both stop at the same tolerance with (nearly) the same preconditioner: the iterates agree far below the tolerance
    rel = ((out.x - ref.x).norm(dim=-2) / ref.x.norm(dim=-2)).max().item()
    assert rel < (2e-5 if out.iterations == ref.iterations else 2e-3), rel
    # and both solve the system: true residual at the level the stopping rule promises
    for r in (out, ref):
        res = (K.matvec(desc, r.x) - rhs).norm(dim=-2) / rhs.norm(dim=-2)
        assert res.max().item() < 5e-3
    with mock.patch.dict(os.environ, {"LO_NO_KRON_ROOT": "1"}):
        off, names_off = _prof_names(lambda: K.cg_solve(desc, rhs, precond=pre_k, tolerance=1e-3))
    assert "precond_fused_kron" not in names_off
    assert torch.equal(off.x, ref.x)  # the switch restores the Q-form path bit for bit


def test_kron_root_iteration_pinned_against_the_reference():
    """Golden g22 at 128 (x) 128 (the real reference's iterate after exactly the iteration count it needed at tolerance
    1e-3; tests/golden/make_golden.py): the Kronecker root form is the path taken and the solution stays within 1e-4
    per column of the reference's."""
    g = load_golden("g22_kron_iteration_pinned")
    K1, K2, sig, rhs = cases.kron_factors(2201, 2, 128, 128, 1)
    its = int(g["iterations_n128"])
    d = torch.from_numpy(np.ascontiguousarray(sig[:, 0])).cuda()
    desc = K.kron_diag_descriptor(torch.from_numpy(K1).cuda(), torch.from_numpy(K2).cuda(), d, const_diag=True)
    L, perm = K.pivoted_cholesky(desc.without_diag(), 15, contiguous=False)
    pre = K.precond_build(L, d, True, perm=perm, kron=desc)
    assert pre.kron is not None, pre.kron_kappa
    res, names = _prof_names(lambda: K.cg_solve(desc, torch.from_numpy(rhs).cuda(), precond=pre, tolerance=0.0,
                                                max_iter=its))
    assert "precond_fused_kron" in names, sorted(names)
    assert res.iterations == its and not res.tolerance_reached
    assert max_rel_err_cols(res.x.cpu().numpy(), g["x_pinned_n128"]) < TOL


def test_kron_root_is_refused_for_nearly_dependent_pivot_rows():
    """Smooth RBF factors: the pivot rows are close to dependent, kappa is large, the build keeps the Q form."""
    B, n1, n2 = 3, 128, 128
    K1, K2, rhs = _spd_factors(7, B, n1, n2, kind="rbf")
    sig = torch.full((B,), 1e-2, device="cuda")
    desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
    L, perm = K.pivoted_cholesky(desc.without_diag(), 15, contiguous=False)
    pre = K.precond_build(L, sig, True, perm=perm, kron=desc)
    assert pre.kron_kappa is not None
    if pre.kron_kappa >= K.KRON_ROOT_MAX_KAPPA:
        assert pre.kron is None
    out, names = _prof_names(lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-3, max_iter=300))
    assert ("precond_fused_kron" in names) == (pre.kron is not None)
    ref = K.cg_solve(desc, rhs, precond=K.precond_build(L, sig, True), tolerance=1e-3, max_iter=300)
    assert abs(out.iterations - ref.iterations) <= 2


def test_operator_api_builds_and_uses_the_kron_root():
    B, n1, n2 = 4, 128, 128
    K1, K2, rhs = _spd_factors(11, B, n1, n2)
    sig = torch.full((B, 1), 1e-2, device="cuda")

    def solve():
        clear_preconditioner_memo()
        A = AddedDiagLinearOperator(KroneckerProductLinearOperator(DenseLinearOperator(K1), DenseLinearOperator(K2)),
                                    ConstantDiagLinearOperator(sig, n1 * n2))
        with settings.cg_tolerance(1e-3), settings.max_cholesky_size(0):
            return A.solve(rhs)

    x, names = _prof_names(solve)
    assert "precond_fused_kron" in names
    with mock.patch.dict(os.environ, {"LO_NO_KRON_ROOT": "1"}):
        x0, names0 = _prof_names(solve)
    assert "precond_fused_kron" not in names0
    rel = ((x - x0).norm(dim=-2) / x0.norm(dim=-2)).max().item()
    assert rel < 1e-4, rel


@pytest.mark.parametrize("B,n1,n2", [(100, 128, 128), (50, 256, 256), (40, 128, 384), (34, 256, 384)])
def test_fused_kron_matvec_matches_two_launches_and_fp64(B, n1, n2):
    g = torch.Generator(device="cuda")
    g.manual_seed(5 + n1 + n2)
    K1 = torch.randn(B, n1, n1, generator=g, device="cuda") / n1 ** 0.5
    K2 = torch.randn(B, n2, n2, generator=g, device="cuda") / n2 ** 0.5
    v = torch.randn(B, n1 * n2, 1, generator=g, device="cuda")
    d_full = torch.rand(B, n1 * n2, generator=g, device="cuda") + 0.5
    V = v.reshape(B, n1, n2).double()
    core = (K1.double() @ V @ K2.double().mT).reshape(B, -1, 1)
    for desc, ref in ((K.kron_diag_descriptor(K1, K2, d_full[:, 0].contiguous(), const_diag=True),
                       core + d_full[:, :1, None].double() * v.double()),
                      (K.kron_diag_descriptor(K1, K2, d_full), core + d_full[..., None].double() * v.double()),
                      (K.kron_diag_descriptor(K1, K2, None), core)):
        y, names = _prof_names(lambda: K.matvec(desc, v))
        assert "kron_fused" in names
        with mock.patch.dict(os.environ, {"LO_NO_KRON_FUSED": "1"}):
            y0, names0 = _prof_names(lambda: K.matvec(desc, v))
        assert "kron_fused" not in names0
        e = ((y.double() - ref).norm(dim=-2) / ref.norm(dim=-2)).max().item()
        e0 = ((y0.double() - ref).norm(dim=-2) / ref.norm(dim=-2)).max().item()
        assert e < 2e-6 and e < 2.0 * e0 + 1e-7, (e, e0)
    # the dot partials of the fused epilogue feed alpha: a CG run is the end-to-end check of that layout
    sig = torch.full((B,), 0.5, device="cuda")
    K1s = K1 @ K1.mT + 0.1 * torch.eye(n1, device="cuda")
    K2s = K2 @ K2.mT + 0.1 * torch.eye(n2, device="cuda")
    desc = K.kron_diag_descriptor(K1s, K2s, sig, const_diag=True)
    out = K.cg_solve(desc, v, tolerance=1e-3)
    with mock.patch.dict(os.environ, {"LO_NO_KRON_FUSED": "1"}):
        ref = K.cg_solve(desc, v, tolerance=1e-3)
    assert out.iterations == ref.iterations
    assert ((out.x - ref.x).norm(dim=-2) / ref.x.norm(dim=-2)).max().item() < 1e-5
