"""Engine selection and the result-only / repeat-with-state protocol of lo_cg_solve_f32 (reference: utils/linear_cg.py:98-359):
the exported plan is what the solver executes, the w-recurrence mode against the three-pass kernel, the oracle and the exact
solution, the R-space pass repeated with the state when the floor is not enough, remembered speculation misses."""
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


class ProbedAddedDiag(AddedDiagLinearOperator):
    _probes = None

    def _probe_vectors_and_norms(self):  # hook: reference _linear_operator.py:629-633
        return self._probes


def _precond(desc, d_t, const=False):
    L, perm = K.pivoted_cholesky(desc, 15)
    if desc.kind == K._hip.LO_OP_LOWRANK_DIAG and desc.R <= 32:
        return K.precond_build(L, d_t, constant_diag=const, root=desc.A0, perm=perm)
    return K.precond_build(L, d_t, constant_diag=const)


# ---------------------------------------------------------------- plan == execution
@pytest.mark.parametrize("kind,N,R,c,nt,form", [
    ("low", 8192, 32, 1, 0, "root+q"), ("low", 8192, 32, 17, 16, "root+q"), ("low", 8192, 32, 16, 16, "q"),
    ("low", 8192, 32, 1, 0, "q"), ("low", 8192, 32, 3, 2, "root"), ("low", 3000, 16, 5, 4, "root+q"),
    ("low", 1500, 8, 9, 0, None), ("low", 1024, 32, 1, 0, None), ("low", 16384, 32, 2, 0, "root+q"),
    ("low", 40000, 32, 1, 0, "q"), ("low", 70000, 8, 1, 0, "q"), ("kron", 16384, 128, 1, 0, "q"),
    ("dense", 2048, 0, 5, 4, "q"), ("dense", 9000, 0, 1, 0, "q"),
])
def test_the_plan_is_what_the_solver_executes(kind, N, R, c, nt, form):
    """`lo_cg_plan_f32` (the pure selection function behind tests/test_host_api.py's table) against what
    `lo_cg_solve_f32` launched on this device (`lo_cg_last_executed`): same engines, same column split, same groups."""
    B = 6 if N >= 16384 or kind == "dense" else 24
    gen = torch.Generator(device="cuda").manual_seed(N + 7 * c)
    if kind == "low":
        Cm = torch.randn(B, N, R, generator=gen, device="cuda") / R ** 0.5
        d = torch.rand(B, N, generator=gen, device="cuda") + 0.5
        desc = K.lowrank_diag_descriptor(Cm, d)
        const = False
    elif kind == "kron":
        n = int(N ** 0.5)
        X1, X2 = (torch.randn(B, n, n, generator=gen, device="cuda") / n ** 0.5 for _ in range(2))
        eye = 0.1 * torch.eye(n, device="cuda")
        d = torch.full((B,), 1e-2, device="cuda")
        desc = K.kron_diag_descriptor(X1 @ X1.mT + eye, X2 @ X2.mT + eye, d, const_diag=True)
        const = True
    else:
        X = torch.randn(B, N, 64, generator=gen, device="cuda") / 8
        d = torch.rand(B, N, generator=gen, device="cuda") + 0.5
        desc = K.dense_diag_descriptor((X @ X.mT).contiguous(), d)
        const = False
    pre = None
    if form is not None:
        L, perm = K.pivoted_cholesky(desc.without_diag() if kind == "kron" else desc, 15, contiguous=(kind != "kron"))
        if kind == "low" and "root" in form:
            pre = K.precond_build(L, d, constant_diag=const, root=desc.A0, perm=perm, need_q=("q" in form))
        elif kind == "kron":
            pre = K.precond_build(L, d, True, perm=perm, kron=desc)
        else:
            pre = K.precond_build(L, d, constant_diag=const)
    rhs = torch.randn(B, N, c, generator=gen, device="cuda")
    if nt:
        rhs[..., :nt] /= rhs[..., :nt].norm(dim=-2, keepdim=True)
    plan = K.cg_plan(desc, c, precond=pre, n_tridiag=nt, max_iter=400)
    res = K.cg_solve(desc, rhs, precond=pre, n_tridiag=nt, tolerance=1e-3, max_iter=400)
    ran = K.cg_last_executed()
    assert res.tolerance_reached
    assert not plan["needs_q"]
    keys = ("resident", "lockstep_cols", "lockstep_group", "serial_engine", "serial_group", "streaming_precond",
            "poll_chunk", "first_stop_iteration")
    if plan["rspace"] == "cols" and ran["streaming_iterations"] == 0:
        # (round 5) all columns on R + 1 coordinates in three streaming launches, and the stop rule held at the floor:
        # lockstep_cols / serial_engine of the PLAN name the engines of a repeat with the state, which did not happen
        assert ran["rspace"] == "cols" and ran["lockstep_cols"] == 0 and ran["serial_engine"] == "none" and ran["lean"]
        keys = ("resident", "streaming_precond", "poll_chunk", "first_stop_iteration")
    else:
        assert ran["rspace"] == (plan["rspace"] if ran["streaming_iterations"] == 0 else "none"), (plan, ran)
    for key in keys:
        assert plan[key] == ran[key], (key, plan, ran)
    if plan["resident"]:
        assert ran["resident_iterations"] == plan["resident_iterations"]
        # (stop at the floor: the lean first pass stands; otherwise it was repeated with the state and CG went on)
        assert ran["lean"] == (plan["lean"] and ran["streaming_iterations"] == 0)
        assert res.iterations == plan["resident_iterations"] + ran["streaming_iterations"] or ran["streaming_iterations"] > 0
    else:
        assert ran["streaming_iterations"] >= res.iterations


# ---------------------------------------------------------------- w by recurrence (k_cg_onchip5 MODE 2)
def _woodbury_exact(C, d, rhs):
    C64, d64, r64 = (torch.from_numpy(a).double().cuda() for a in (C, d, rhs))
    Cd = C64 / d64.unsqueeze(-1)
    cap = torch.eye(C64.shape[-1], dtype=torch.float64, device="cuda") + C64.mT @ Cd
    return (r64 / d64.unsqueeze(-1) - Cd @ torch.linalg.solve(cap, C64.mT @ (r64 / d64.unsqueeze(-1)))).cpu().numpy()

@pytest.mark.parametrize("N,R,B,dscale,doff,cscale", [
    (8192, 32, 40, 1.0, 0.5, 1.0),      # the headline spectrum
    (8192, 32, 24, 0.1, 0.01, 1.0),     # small diagonals
    (2048, 32, 24, 0.01, 0.001, 1.0),
    (4096, 16, 24, 1.0, 0.05, 1.0),
    (5000, 8, 24, 1.0, 0.5, 1.0),       # rank <= pivots: P = A
    (8192, 32, 24, 1.0, 0.5, 10.0),     # strong low-rank part
    (16384, 32, 12, 1.0, 0.5, 1.0),     # groups of 16
    (1024, 32, 24, 10.0, 0.5, 1.0),     # a group of one
    (40000, 32, 6, 1.0, 0.5, 1.0),      # groups of 64
])
def test_w_recurrence_mode_against_three_pass_oracle_and_exact_solution(N, R, B, dscale, doff, cscale, monkeypatch):
    """Single-column solves without tridiagonals carry w = C^T D^-1 r by recurrence (k_cg_onchip5 MODE 2; numerics
    prototype tests/proto/proto_w_recurrence.py; selected with LO_OC_NO_RSPACE since round 5).  Same iteration count as
    the three-pass iteration (LO_OC_NO_WREC) and the oracle; solution within 1e-4 per column of the oracle's
    (north_star's bar) and as close to the EXACT solution (fp64 Woodbury) as the three-pass iteration is, within a
    factor of 3."""
    C, d, rhs = cases.lowrank_diag(8800 + R, B, N, R, 1)
    C = (C * cscale).astype(np.float32)
    d = ((d - 0.5) * dscale + doff).astype(np.float32)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    L, perm = K.pivoted_cholesky(desc, 15)
    pre = K.precond_build(L, dev(d), constant_diag=False, root=desc.A0, perm=perm)
    assert pre.E is not None
    monkeypatch.setenv("LO_OC_NO_RSPACE", "1")
    K._hip.prof_enable(True)
    res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    torch.cuda.synchronize()
    prof = K._hip.prof_report()
    K._hip.prof_enable(False)
    assert "cg_onchip" in prof and K.cg_last_executed()["serial_engine"] == "root"
    ran_lean = K.cg_last_executed()["lean"]  # the result-only (w-recurrence) pass met the stop rule and stands
    monkeypatch.setenv("LO_OC_NO_WREC", "1")
    ref = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    monkeypatch.delenv("LO_OC_NO_WREC")
    assert res.iterations == ref.iterations and res.tolerance_reached == ref.tolerance_reached
    if ran_lean and R > 15:
        # (stop at the floor: the w-recurrence result stands.  A root of rank <= 15 is reproduced exactly by the pivots:
        #  P = A, CG converges in its first step -- before the recurrence has been used at all)
        assert not torch.equal(res.x, ref.x), "the two modes gave identical bits: the switch does not switch"
    exact = _woodbury_exact(C, d, rhs)
    e_wr, e_3p = max_rel_err_cols(host(res.x), exact), max_rel_err_cols(host(ref.x), exact)
    # (the ill-conditioned cases miss the stop rule at the floor: the w-recurrence pass is discarded, the three-pass kernel
    #  repeats it and CG continues -- the very same bits as the reference run; their distance from the EXACT solution is
    #  tolerance x condition number, for both)
    assert e_wr < 3 * e_3p + 1e-6 and (e_wr < 1e-4 or e_wr == e_3p), (e_wr, e_3p)
    sub = slice(0, 3)
    Lo, _ = orc.pivoted_cholesky(orc.LowRankRowSource(C[sub]), 15)
    xo, _, info = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C[sub], d[sub], v), rhs[sub], tolerance=1e-4,
                                preconditioner=orc.Preconditioner(Lo, d[sub]).apply)
    assert max_rel_err_cols(host(res.x)[sub], xo) < (1e-4 if e_3p < 1e-5 else 2e-3)
    res2 = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    assert torch.equal(res.x, res2.x)  # bitwise reproducible
    monkeypatch.delenv("LO_OC_NO_RSPACE")

    # ---- round 5: the default result-only pass runs the iterations on R + 1 coordinates in fp64 (k_cg_rspace, numerics
    # prototype tests/proto/proto_rspace.py): the same CG in exact arithmetic.  Where the fp32 iterations stop at the floor
    # it reports the same iteration count; where their rounding makes them go on (ill-conditioned members) it stops at
    # the floor with a solution that is CLOSER to the exact one.
    assert pre.RS is not None
    K.set_onchip_cg(True)  # (forgets the miss of the w-recurrence pass above: speculate afresh)
    rs = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    ran = K.cg_last_executed()
    assert ran["serial_engine"] == "root" and ran["lean"] and rs.tolerance_reached
    if ran_lean:
        assert rs.iterations == ref.iterations
        assert not torch.equal(rs.x, res.x), "the R-space kernel did not run"
    else:
        assert rs.iterations <= ref.iterations
    e_rs = max_rel_err_cols(host(rs.x), exact)
    assert e_rs < 2e-6 and e_rs < 3 * e_3p + 1e-6, (e_rs, e_3p)
    assert max_rel_err_cols(host(rs.x)[sub], xo) < (1e-4 if e_3p < 1e-5 else 2e-3)
    rs2 = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    assert torch.equal(rs.x, rs2.x)  # bitwise reproducible


def test_rspace_pass_is_repeated_with_the_state_when_the_floor_is_not_enough():
    """Round 5.  Columns of C with geometrically decaying norms (32 well separated eigenvalue clusters), a rank-2
    preconditioner and a tolerance the 11 iterations of the floor cannot meet even in exact arithmetic: the result-only
    R-space pass (one column: k_cg_rspace; three columns: k_rs_part / k_rs_iter / k_rs_apply) misses the stop rule, the
    three-pass resident kernel repeats the iterations WITH the state and the streaming engine continues from it."""
    B, N, R = 12, 8192, 32
    for c in (1, 3):
        C, d, rhs = cases.lowrank_diag(8890 + c, B, N, R, c)
        C = (C * (0.8 ** np.arange(R))[None, None, :]).astype(np.float32)
        desc = K.lowrank_diag_descriptor(dev(C), dev(d))
        L, perm = K.pivoted_cholesky(desc, 2)
        pre = K.precond_build(L, dev(d), constant_diag=False, root=desc.A0, perm=perm)
        assert pre.RS is not None
        K.set_onchip_cg(True)
        plan = K.cg_plan(desc, c, precond=pre, max_iter=200)
        assert plan["rspace"] == ("resident" if c == 1 else "cols") and plan["lean"]
        K._hip.prof_enable(True)
        res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-5, max_iter=200)
        torch.cuda.synchronize()
        prof = K._hip.prof_report()
        K._hip.prof_enable(False)
        ran = K.cg_last_executed()
        assert ("rs_iter" in prof) == (c > 1) and "cg_onchip" in prof  # the R-space pass ran, then the repeat
        assert ran["resident"] and ran["streaming_iterations"] > 0 and not ran["lean"] and ran["rspace"] == "none"
        assert res.tolerance_reached and res.iterations > 11
        exact = np.concatenate([_woodbury_exact(C, d, rhs[..., j:j + 1]) for j in range(c)], -1)
        assert max_rel_err_cols(host(res.x), exact) < 1e-4


@pytest.mark.usefixtures("legacy_resident_engines")
def test_w_recurrence_mode_continues_on_the_streaming_engine_when_the_floor_is_not_enough():
    """Tolerance far below what 11 iterations reach with a weak (rank-2) preconditioner: the result-only first pass (w by
    recurrence) misses the stop rule, is repeated by the three-pass kernel WITH the state, and the streaming engine
    continues from its x / r / p / z."""
    B, N, R = 12, 8192, 32
    C, d, rhs = cases.lowrank_diag(8899, B, N, R, 1)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    L, perm = K.pivoted_cholesky(desc, 2)
    pre = K.precond_build(L, dev(d), constant_diag=False, root=desc.A0, perm=perm)
    res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-6, max_iter=200)
    ran = K.cg_last_executed()
    assert ran["resident"] and ran["streaming_iterations"] > 0 and res.tolerance_reached
    assert max_rel_err_cols(host(res.x), _woodbury_exact(C, d, rhs)) < 1e-4


@pytest.mark.usefixtures("legacy_resident_engines")
def test_speculation_misses_are_remembered(monkeypatch):
    """(i) a solve whose result-only first pass misses the stop rule at the floor starts its NEXT solve with
    the state-writing pass (one resident launch instead of two); a solve that stops at the floor again clears the entry.
    (ii) The one-launch `A.solve` that had to be redone by the three-launch path is not tried again for the next solves
    of the same operator."""
    B, N, R = 12, 8192, 32
    C, d, rhs = cases.lowrank_diag(8899, B, N, R, 1)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    L, perm = K.pivoted_cholesky(desc, 2)
    pre = K.precond_build(L, dev(d), constant_diag=False, root=desc.A0, perm=perm)

    def launches(tol):
        K._hip.prof_enable(True)
        res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=tol, max_iter=200)
        torch.cuda.synchronize()
        prof = K._hip.prof_report()
        K._hip.prof_enable(False)
        return res, prof["cg_onchip"][0]

    K.set_onchip_cg(True)     # (forgets earlier misses: the memo is keyed on the operator's device pointers)
    r1, n1 = launches(1e-6)   # weak preconditioner, tolerance beyond the floor: result-only pass + repeat with the state
    r2, n2 = launches(1e-6)   # remembered: the state-writing pass at once
    assert (n1, n2) == (2, 1) and r1.iterations == r2.iterations and torch.equal(r1.x, r2.x)
    r3, n3 = launches(1.0)    # another tolerance is another signature: speculation as usual, holds at the floor
    assert n3 == 1 and r3.iterations == 11 and K.cg_last_executed()["lean"]
    # (ii) the operator API: a hard system (rank-2 preconditioner cannot meet 1e-6 at the floor)
    import sys

    import linear_operator_amd.utils  # noqa: F401
    lcg = sys.modules["linear_operator_amd.utils.linear_cg"]  # (the attribute of the same name is the function)
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo

    lcg._FUSED_MISSES.clear()
    calls = []
    real = K.solve_fused
    monkeypatch.setattr(K, "solve_fused", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    import warnings
    with settings.cg_tolerance(1e-7), settings.max_preconditioner_size(2), settings.max_cg_iterations(60), \
            warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xs = []
        C_t, d_t, rhs_t = desc.A0.reshape(B, N, R), dev(d), dev(rhs)  # (the operator's tensors persist, as in a training loop)
        for _ in range(3):
            clear_preconditioner_memo()
            A = AddedDiagLinearOperator(LowRankRootLinearOperator(C_t), DiagLinearOperator(d_t))
            xs.append(A.solve(rhs_t))
    assert len(calls) == 1, f"the one-launch solve was speculated {len(calls)} times for an operator that had missed"
    assert torch.equal(xs[0], xs[1]) and torch.equal(xs[1], xs[2])
    lcg._FUSED_MISSES.clear()
