"""GPU tests of the ONE-LAUNCH end-to-end solve (lo_solve_fused_f32, csrc/lo_solve_fused_impl.h): pivoted Cholesky ->
root-form preconditioner -> preconditioned CG of AddedDiag(LowRankRoot, Diag | ConstantDiag) in one resident kernel.

Checked against (i) the golden vectors the real reference produced (pivots exactly, solutions <= 1e-4), (ii) the oracle
on the same seeded inputs (pivots bit-exact, solution <= 1e-4 per column, logdet P), (iii) the three-launch HIP path
(same pivots, same root form to fp32 rounding), and through the host API (A.solve / torch.linalg.solve, the lazy
preconditioner closure, its memo, the fallbacks)."""
import warnings
from unittest import mock

import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, max_rel_err_cols

pytestmark = pytest.mark.gpu

import linear_operator_amd as lo  # noqa: E402
from linear_operator_amd import kernels as K  # noqa: E402
from linear_operator_amd import settings  # noqa: E402
from linear_operator_amd.operators import (  # noqa: E402
    AddedDiagLinearOperator, ConstantDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator,
)
from linear_operator_amd.operators import added_diag_linear_operator as adl  # noqa: E402
from linear_operator_amd.utils.warnings import NumericalWarning  # noqa: E402
from oracle import lo_oracle as orc  # noqa: E402  (the checker)

TOL = 1e-4  # north_star bar on fp32 solves, relative per column


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


def _oracle_solve(C, d, rhs, rank=15, tol=1e-4):
    L, piv = orc.pivoted_cholesky(orc.LowRankRowSource(C), rank)
    pre = orc.Preconditioner(L, d)
    x, _, info = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, tolerance=tol, preconditioner=pre.apply)
    return x, piv, pre, info


@pytest.mark.parametrize("N,R,B,c", [(8192, 32, 70, 1), (2048, 16, 20, 2), (1000, 8, 37, 1), (5000, 32, 9, 3),
                                     (4096, 32, 130, 1), (300, 16, 600, 1), (8192, 16, 9, 1), (7000, 8, 11, 4),
                                     (16384, 32, 40, 1), (12000, 32, 7, 2), (9000, 16, 5, 1)])
def test_fused_solve_matches_oracle_pivots_and_solution(N, R, B, c):
    """Members larger / smaller than a workgroup multiple, every group size (1, 2, 4, 8, 16 workgroups), 1-4 columns:
    pivots bit-exact vs the oracle, solution within 1e-4 per column, logdet P of the root form, iteration floor."""
    C, d, rhs = cases.lowrank_diag(7100 + R + N, B, N, R, c)
    if R == 8:  # a rank-8 root exhausts after 8 pivots (the batch-global rule stops early): factor a rank-12 request
        rank = 8
    else:
        rank = 15
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    assert K.solve_fused_supported(desc, c, rank)
    K._hip.prof_enable(True)
    rf = K.solve_fused(desc, dev(rhs), rank, 1e-3, tolerance=TOL)
    torch.cuda.synchronize()
    prof = K._hip.prof_report()
    K._hip.prof_enable(False)
    assert rf is not None and list(prof) == ["solve_fused"], f"one launch expected, ran {list(prof)}"
    assert rf.cg.iterations == 11 and rf.cg.tolerance_reached and not rf.cg.nan_detected
    sub = slice(0, 3)
    xo, piv, pre, info = _oracle_solve(C[sub], d[sub], rhs[sub], rank)
    assert info.iterations == 11
    assert np.array_equal(host(rf.permutation(N))[sub], piv), "pivots differ from the oracle"
    assert max_rel_err_cols(host(rf.cg.x)[sub], xo) < TOL
    assert np.allclose(host(rf.precond.logdet).reshape(-1)[sub], pre.logdet, rtol=1e-5, atol=1e-3)
    # exact solution of the same systems (fp64 Woodbury closed form), all members
    xs = orc.woodbury_solve(C.astype(np.float64), d.astype(np.float64), rhs.astype(np.float64))
    assert max_rel_err_cols(host(rf.cg.x), xs) < TOL
    # the three-launch path on the same inputs: same pivots everywhere, same root form up to fp32 rounding
    L, perm = K.pivoted_cholesky(desc, rank, contiguous=False)
    pre3 = K.precond_build(L, dev(d), False, root=dev(C), perm=perm, need_q=False)
    r3 = K.cg_solve(desc, dev(rhs), precond=pre3, tolerance=TOL)
    assert torch.equal(rf.permutation(N), perm.reshape(B, N))
    assert max_rel_err_cols(host(rf.cg.x), host(r3.x)) < 2e-5
    for a, b in ((rf.precond.F, pre3.F), (rf.precond.EF, pre3.EF), (rf.precond.E, pre3.E)):
        assert float((a - b).norm() / b.norm()) < 1e-5
    assert np.allclose(host(rf.precond.logdet).reshape(-1), host(pre3.logdet).reshape(-1), rtol=1e-6, atol=1e-3)
    assert np.allclose(host(rf.precond.dinv), host(pre3.dinv), rtol=3e-7)
    # later solves with the preconditioner the fused launch built: the root-form CG kernel alone
    again = K.cg_solve(desc, dev(rhs), precond=rf.precond, tolerance=TOL)
    assert max_rel_err_cols(host(again.x), host(rf.cg.x)) < 2e-5


def test_fused_solve_against_the_reference_goldens():
    """g4_solve_lowrank: the real reference's A.solve (CG + rank-15 pivoted-Cholesky preconditioner) and the exact
    solution; g2_pivchol_lowrank: its pivots.  The rank-8 root of g2 exhausts after 8 pivots -- the reference stops the
    whole batch there (_pivoted_cholesky.py:57) -- so the fused launch must DECLINE (status EARLY_STOP)."""
    g = load_golden("g4_solve_lowrank")
    C, d, rhs = cases.lowrank_diag(401, 4, 2048, 32, 1)
    rf = K.solve_fused(K.lowrank_diag_descriptor(dev(C), dev(d)), dev(rhs), 15, 1e-3, tolerance=TOL)
    assert rf is not None and rf.cg.iterations + 1 == int(g["matvecs"])  # (the reference also counts A x0)
    assert max_rel_err_cols(host(rf.cg.x), g["x"]) < TOL and max_rel_err_cols(host(rf.cg.x), g["x_exact"]) < TOL
    g2 = load_golden("g2_pivchol_lowrank")
    C32, d32, rhs32 = cases.lowrank_diag(210 + 32, 3, 2048, 32, 1)
    rf = K.solve_fused(K.lowrank_diag_descriptor(dev(C32), dev(d32)), dev(rhs32), 15, 1e-3, tolerance=TOL)
    assert rf is not None and np.array_equal(host(rf.permutation(2048)), g2["piv_R32"])
    C8, d8, rhs8 = cases.lowrank_diag(210 + 8, 3, 2048, 8, 1)
    assert g2["L_R8"].shape[-1] == 8  # the reference took 8 pivots of the 15 requested
    assert K.solve_fused(K.lowrank_diag_descriptor(dev(C8), dev(d8)), dev(rhs8), 15, 1e-3, tolerance=TOL) is None


def test_fused_solve_constant_diagonal_and_zero_columns():
    C, _, rhs = cases.lowrank_diag(7301, 12, 3000, 16, 2)
    sig = (np.random.default_rng(5).random(12) + 0.5).astype(np.float32)
    rhs[3, :, 1] = 0.0  # a zero right-hand side column returns exactly zero (linear_cg.py:177-179, :299)
    desc = K.lowrank_diag_descriptor(dev(C), dev(sig), const_diag=True)
    rf = K.solve_fused(desc, dev(rhs), 15, 1e-3, tolerance=TOL)
    assert rf is not None and rf.precond.constant_diag and tuple(rf.precond.dinv.shape) == (12,)
    dfull = np.repeat(sig[:, None], 3000, axis=1)
    xs = orc.woodbury_solve(C.astype(np.float64), dfull.astype(np.float64), rhs.astype(np.float64))
    x = host(rf.cg.x)
    assert np.all(x[3, :, 1] == 0.0)
    keep = np.ones((12, 2), bool)
    keep[3, 1] = False
    num = np.sqrt(((x - xs) ** 2).sum(-2))[keep]
    den = np.sqrt((xs ** 2).sum(-2))[keep]
    assert float((num / den).max()) < TOL
    xo, piv, pre, _ = _oracle_solve(C[:2], dfull[:2], rhs[:2])
    assert np.array_equal(host(rf.permutation(3000))[:2], piv)
    assert np.allclose(host(rf.precond.logdet).reshape(-1)[:2], pre.logdet, rtol=1e-5, atol=1e-3)


def test_fused_solve_declines_what_it_cannot_decide(monkeypatch):
    C, d, rhs = cases.lowrank_diag(7401, 6, 2048, 32, 1)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    # tolerance below what fp32 reaches at the floor: CG has to continue -> three-launch path
    assert K.solve_fused(desc, dev(rhs), 15, 1e-3, tolerance=1e-9) is None
    # a loose pivot tolerance (the error is ||remaining diagonal||_1 / max diagonal ~ N): members stop before 15 pivots
    # -> the batch-global rule is needed
    assert K.solve_fused(desc, dev(rhs), 15, 1500.0, tolerance=TOL) is None
    # a timed-out group exchange
    monkeypatch.setenv("LO_OC_TEST_FALLBACK", "1")
    assert K.solve_fused(desc, dev(rhs), 15, 1e-3, tolerance=TOL) is None
    monkeypatch.delenv("LO_OC_TEST_FALLBACK")
    assert K.solve_fused(desc, dev(rhs), 15, 1e-3, tolerance=TOL) is not None
    # shapes outside the kernel: ranks above 16, roots that are not 8 / 16 / 32 wide, members above 16384 rows
    assert not K.solve_fused_supported(desc, 1, 17) and not K.solve_fused_supported(desc, 9, 15)
    C2 = cases.lowrank_diag(7402, 2, 2048, 20, 1)[0]
    assert not K.solve_fused_supported(K.lowrank_diag_descriptor(dev(C2), dev(d[:2])), 1, 15)
    C3, d3, _ = cases.lowrank_diag(7403, 2, 17000, 16, 1)
    assert not K.solve_fused_supported(K.lowrank_diag_descriptor(dev(C3), dev(d3)), 1, 15)
    assert K.solve_fused_supported(K.lowrank_diag_descriptor(dev(C3[:, :9000]), dev(d3[:, :9000])), 1, 15)  # (round 4)
    # NaN in the root: flagged, never a silent result
    Cn = C.copy()
    Cn[2, 17, 3] = np.nan
    assert K.solve_fused(K.lowrank_diag_descriptor(dev(Cn), dev(d)), dev(rhs), 15, 1e-3, tolerance=TOL) is None


def _three_launch_solve(A, rhs, monkeypatch):
    monkeypatch.setenv("LO_NO_FUSED_SOLVE", "1")
    adl.clear_preconditioner_memo()
    try:
        return A.solve(rhs)
    finally:
        monkeypatch.delenv("LO_NO_FUSED_SOLVE")
        adl.clear_preconditioner_memo()


def test_operator_api_solve_takes_the_fused_launch(monkeypatch):
    """A.solve / torch.linalg.solve of AddedDiag(LowRankRoot, Diag): `_solve_preconditioner` defers the factorisation,
    `utils.linear_cg` (still the patchable seam) runs ONE kernel; a second solve with the same tensors reuses the
    memoised root form; inv_quad_logdet afterwards builds the full preconditioner (probes need L)."""
    C, d, rhs = cases.lowrank_diag(7501, 40, 4096, 32, 1)
    Ct, dt, rt = dev(C), dev(d), dev(rhs)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
    adl.clear_preconditioner_memo()
    spy_cg = mock.MagicMock(wraps=lo.utils.linear_cg)
    spy_fused = mock.MagicMock(wraps=K.solve_fused)
    spy_pc = mock.MagicMock(wraps=K.pivoted_cholesky)
    with settings.cg_tolerance(TOL), mock.patch("linear_operator_amd.utils.linear_cg", new=spy_cg), \
            mock.patch.object(K, "solve_fused", new=spy_fused), mock.patch.object(K, "pivoted_cholesky", new=spy_pc):
        K._hip.prof_enable(True)
        x = torch.linalg.solve(A, rt)
        torch.cuda.synchronize()
        prof = K._hip.prof_report()
        K._hip.prof_enable(False)
        assert spy_cg.call_count == 1 and spy_fused.call_count == 1 and spy_pc.call_count == 0
        assert isinstance(spy_cg.call_args.kwargs["preconditioner"], adl.LazyWoodburyPreconditionClosure)
        assert list(prof) == ["solve_fused"]
        # second solve, same tensors: the memoised root form, CG kernel only
        K._hip.prof_enable(True)
        x2 = A.solve(rt)
        torch.cuda.synchronize()
        prof2 = K._hip.prof_report()
        K._hip.prof_enable(False)
        assert spy_fused.call_count == 1 and spy_pc.call_count == 0 and list(prof2) == ["cg_onchip"]
    xs = orc.woodbury_solve(C.astype(np.float64), d.astype(np.float64), rhs.astype(np.float64))
    assert max_rel_err_cols(host(x), xs) < TOL and max_rel_err_cols(host(x2), xs) < TOL
    with settings.cg_tolerance(TOL):
        x3 = _three_launch_solve(A, rt, monkeypatch)
    assert max_rel_err_cols(host(x), host(x3)) < 2e-5
    # logdet after a fused solve: the root-form memo is not enough (probes are sampled from L): full build, and the same
    # value as without any fused solve before it (same seed -> same probes from the same factor)
    def iql():
        torch.manual_seed(0)
        with settings.cg_tolerance(TOL), settings.num_trace_samples(16):
            return A.inv_quad_logdet(rt, logdet=True)

    iq, ld = iql()
    monkeypatch.setenv("LO_NO_FUSED_SOLVE", "1")
    adl.clear_preconditioner_memo()
    iq3, ld3 = iql()
    monkeypatch.delenv("LO_NO_FUSED_SOLVE")
    assert np.allclose(host(ld), host(ld3), rtol=1e-4, atol=1e-2) and np.allclose(host(iq), host(iq3), rtol=1e-4)
    exact = orc.woodbury_logdet(C.astype(np.float64), d.astype(np.float64))
    assert np.abs(host(ld) - exact).max() < 40.0  # (stochastic estimate, 16 probes: sanity only)
    assert np.allclose(host(iq), (xs * rhs).sum(-2).sum(-1), rtol=1e-3)
    adl.clear_preconditioner_memo()


def test_lazy_preconditioner_closure_is_a_drop_in(monkeypatch):
    """The deferred closure behaves like the reference's precondition_closure whenever it is USED as one: calling it,
    a CG call the fused kernel does not take (tridiagonals, initial guess, another operator's matmul), warnings."""
    C, d, rhs = cases.lowrank_diag(7601, 5, 2048, 16, 2)
    Ct, dt, rt = dev(C), dev(d), dev(rhs)

    def fresh():  # (the caches live on the operator object: a new object per scenario, as Solve.forward rebuilds it)
        adl.clear_preconditioner_memo()
        return AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))

    A = fresh()
    lazy = A._solve_preconditioner()
    assert isinstance(lazy, adl.LazyWoodburyPreconditionClosure) and lazy.pending
    z = lazy(rt)  # materialises: z = P^-1 r (added_diag_linear_operator.py:135-140)
    assert not lazy.pending
    _, _, pre, _ = _oracle_solve(C, d, rhs[..., :1])
    assert max_rel_err_cols(host(z), pre.apply(rhs)) < 1e-4
    # tridiagonals requested: not the fused kernel's business
    A = fresh()
    lazy = A._solve_preconditioner()
    assert lazy.pending
    with mock.patch.object(K, "solve_fused", new=mock.MagicMock(wraps=K.solve_fused)) as spy:
        x, t = lo.utils.linear_cg(A._matmul, rt, n_tridiag=1, tolerance=TOL, preconditioner=lazy)
        assert spy.call_count == 0 and t.shape[0] == 1 and not lazy.pending
    xs = orc.woodbury_solve(C.astype(np.float64), d.astype(np.float64), rhs.astype(np.float64))
    assert max_rel_err_cols(host(x), xs) < TOL
    # another operator's matmul with this closure: never fused (the closure belongs to A)
    A = fresh()
    lazy = A._solve_preconditioner()
    assert lazy.pending
    B2 = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct.clone()), DiagLinearOperator(dt.clone()))
    with mock.patch.object(K, "solve_fused", new=mock.MagicMock(wraps=K.solve_fused)) as spy:
        x = lo.utils.linear_cg(B2._matmul, rt, tolerance=TOL, preconditioner=lazy)
        assert spy.call_count == 0
    assert max_rel_err_cols(host(x), xs) < TOL
    # tolerance the floor cannot meet: fused declines, the ordinary path continues and warns like the reference
    A = fresh()
    with settings.cg_tolerance(1e-9), settings.max_cg_iterations(30), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x = A.solve(rt)
    assert any(issubclass(m.category, NumericalWarning) for m in w)
    assert max_rel_err_cols(host(x), xs) < TOL
    adl.clear_preconditioner_memo()


def test_fused_solve_gradients_through_the_solve_function(monkeypatch):
    """Solve.backward needs one more solve with the same operator (the memoised root form) and the bilinear
    derivative: gradients equal to the three-launch path's."""
    C, d, rhs = cases.lowrank_diag(7701, 6, 2048, 16, 1)

    def run(fused):
        Cg, dg = dev(C).requires_grad_(True), dev(d).requires_grad_(True)
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
        adl.clear_preconditioner_memo()
        if not fused:
            monkeypatch.setenv("LO_NO_FUSED_SOLVE", "1")
        try:
            with settings.cg_tolerance(TOL):
                x = A.solve(dev(rhs))
                (x * x).sum().backward()
        finally:
            if not fused:
                monkeypatch.delenv("LO_NO_FUSED_SOLVE")
        return host(x), host(Cg.grad), host(dg.grad)

    xf, gCf, gdf = run(True)
    x3, gC3, gd3 = run(False)
    assert max_rel_err_cols(xf, x3) < 2e-5
    assert np.linalg.norm(gCf - gC3) / np.linalg.norm(gC3) < 1e-4
    assert np.linalg.norm(gdf - gd3) / np.linalg.norm(gd3) < 1e-4
    adl.clear_preconditioner_memo()


def test_fused_solve_with_constant_diag_operator_api():
    C, _, rhs = cases.lowrank_diag(7801, 8, 2500, 32, 1)
    sig = torch.full((8, 1), 0.7, device="cuda")
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), ConstantDiagLinearOperator(sig, 2500))
    adl.clear_preconditioner_memo()
    with settings.cg_tolerance(TOL), mock.patch.object(K, "solve_fused", new=mock.MagicMock(wraps=K.solve_fused)) as spy:
        x = A.solve(dev(rhs))
        assert spy.call_count == 1
    dfull = np.full((8, 2500), 0.7)
    xs = orc.woodbury_solve(C.astype(np.float64), dfull, rhs.astype(np.float64))
    assert max_rel_err_cols(host(x), xs) < TOL
    adl.clear_preconditioner_memo()


def test_fused_solve_repeated_runs_are_bit_identical():
    """300 solves through the operator API with allocator churn in between (tools/stress_fused.py in small): the kernel
    is deterministic by construction (fixed summation orders, tagged hand-offs), every result equals the first bit for
    bit and every solve is the one-launch kernel."""
    B, N, R = 64, 8192, 32
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    d = torch.rand(B, N, generator=g, device="cuda") + 0.5
    rhs = torch.randn(B, N, 1, generator=g, device="cuda")
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
    first = None
    with settings.cg_tolerance(1e-4):
        for i in range(300):
            adl.clear_preconditioner_memo()
            if i % 7 == 3:
                junk = [torch.empty(int(s), device="cuda").normal_() for s in (1e5, 3e6, 7e4)]
                del junk
            K._hip.prof_enable(True)
            try:
                x = A.solve(rhs)
                names = set(K._hip.prof_report())
            finally:
                K._hip.prof_enable(False)
            assert names == {"solve_fused"}, (i, sorted(names))
            if first is None:
                first = x.clone()
            assert torch.equal(x, first), (i, float((x - first).abs().max()))
    adl.clear_preconditioner_memo()
