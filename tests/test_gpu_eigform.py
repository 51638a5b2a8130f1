"""Round 5 (second half): the DIAGONAL form of the R-space CG (lo_precond_eigform_f32 + k_cg_rspace<.., true>).

  * the form itself: E E^+ E = E, Tin^T (E + E^2) Tin = Lam, Tu^T (E - E F E) Tu = I on the kept directions, for full-rank,
    low-rank (duplicated columns) and small roots -- the identities the iteration relies on (csrc/lo_eigform.hip);
  * the solve: same iteration count, stop-rule statistics and solutions as the dense R-space kernel (1e-6), within 1e-4 of
    the C oracle's linear_cg (utils/linear_cg.py:245-332) and of the fp64 Woodbury solution;
  * zero right-hand sides and NaN right-hand sides take the reference's branches (:178-179, :199-200);
  * a preconditioner cache gets the form when the single-column solves it has served pay for it (kernels._eigform_due;
    these tests fix the count at one with kernels.EIGFORM_AFTER_USES).
"""
import os
import warnings

import numpy as np
import pytest
import torch

import cases
from conftest import max_rel_err_cols

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from oracle import lo_oracle_c as occ  # noqa: E402  (the checker, C restatement)


@pytest.fixture(autouse=True)
def _form_on_second_use():
    old = K.EIGFORM_AFTER_USES
    K.EIGFORM_AFTER_USES = 1
    yield
    K.EIGFORM_AFTER_USES = old


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


def _setup(seed, B, N, R, dup=False, dscale=1.0, doff=0.5, rank=15):
    C, d, rhs = cases.lowrank_diag(seed, B, N, R, 1)
    if dup:
        C[..., R // 2:] = C[..., :R - R // 2]
    d = ((d - 0.5) * dscale + doff).astype(np.float32)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    L, perm = K.pivoted_cholesky(desc, rank)
    pre = K.precond_build(L, dev(d), constant_diag=False, root=desc.A0, perm=perm)
    assert pre.RS is not None and pre.rs_rank == R
    return C, d, rhs, desc, pre, L


def _woodbury(C, d, rhs):
    C64, d64, r64 = (torch.from_numpy(a).double().cuda() for a in (C, d, rhs))
    Cd = C64 / d64.unsqueeze(-1)
    cap = torch.eye(C64.shape[-1], dtype=torch.float64, device="cuda") + C64.mT @ Cd
    return r64 / d64.unsqueeze(-1) - Cd @ torch.linalg.solve(cap, C64.mT @ (r64 / d64.unsqueeze(-1)))


@pytest.mark.parametrize("N,R,dup,dscale,doff", [(4096, 32, False, 1.0, 0.5), (2048, 32, True, 1.0, 0.5),
                                                  (3000, 16, False, 1.0, 0.05), (1500, 8, False, 1.0, 0.5),
                                                  (2048, 32, False, 0.01, 0.001), (3000, 20, False, 1.0, 0.5)])
def test_eigform_identities(N, R, dup, dscale, doff):
    *_, pre, _L = _setup(7100 + R, 12, N, R, dup=dup, dscale=dscale, doff=doff)
    pre.ensure_eigform()
    assert torch.is_tensor(pre.RSD), "the form must be usable for these members"
    RS, D = pre.RS, pre.RSD
    E, F = RS[:, 0, :R, :R], RS[:, 4, :R, :R]
    TinT, Ep, TuT, Tin = D[:, 0, :R, :R], D[:, 1, :R, :R], D[:, 2, :R, :R], D[:, 4, :R, :R]
    lam = D[:, 5, 0, :R]
    st = D[:, 5, 1, :4]
    assert (st[:, 0] == 1.0).all() and (st[:, 1] < 20).all() and (st[:, 2] < 20).all(), st
    rank = st[:, 3]
    assert (rank == (R // 2 if dup else R)).all(), rank
    sc = E.abs().amax((-1, -2), keepdim=True)
    assert ((E @ Ep @ E - E).abs() / sc).max().item() < 1e-12
    keep = (Tin.abs().amax(-2) > 0).double()
    assert (keep.sum(-1) == rank).all()
    A_u = TinT @ (E + E @ E) @ Tin
    assert ((A_u - torch.diag_embed(lam * keep)).abs().max() / lam.max()).item() < 1e-11
    P_u = TuT @ (E - E @ F @ E) @ TuT.mT
    assert (P_u - torch.diag_embed(keep)).abs().max().item() < 1e-8
    assert torch.equal(TinT, Tin.mT)
    assert (lam >= 1.0 - 1e-6).all(), f"the pivoted-Cholesky preconditioner never exceeds the operator: {lam.min().item()}"
    # padding: zero rows / columns, unit eigenvalues
    ld = D.shape[-1]
    if ld > R:
        assert (D[:, :5, R:, :] == 0).all() and (D[:, :5, :, R:] == 0).all() and (D[:, 5, 0, R:] == 1).all()


@pytest.mark.parametrize("B,N,R,dup,dscale,doff", [(70, 4096, 32, False, 1.0, 0.5), (24, 2048, 32, True, 1.0, 0.5),
                                                    (20, 8192, 32, False, 0.1, 0.01), (16, 5000, 8, False, 1.0, 0.5),
                                                    (12, 3000, 20, False, 1.0, 0.5), (9, 16384, 32, False, 1.0, 0.5)])
def test_diagonal_form_solve_matches_dense_rspace_oracle_and_exact(B, N, R, dup, dscale, doff):
    C, d, rhs, desc, pre, L = _setup(7200 + R + B, B, N, R, dup=dup, dscale=dscale, doff=doff)
    rd = dev(rhs)
    dense = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)  # first use of the cache: the dense R-space iteration
    e0 = K.cg_last_executed()
    assert e0["rspace"] == "resident" and not e0["rspace_diag"], e0
    diag = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)   # second use: the cache gets the diagonal form
    e1 = K.cg_last_executed()
    assert e1["rspace"] == "resident" and e1["rspace_diag"] and e1["lean"], e1
    assert torch.is_tensor(pre.RSD)
    assert diag.iterations == dense.iterations and diag.tolerance_reached == dense.tolerance_reached
    assert abs(diag.mean_residual - dense.mean_residual) <= 1e-5 * dense.mean_residual + 1e-9
    assert max_rel_err_cols(host(diag.x), host(dense.x)) < 1e-6
    again = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
    assert torch.equal(again.x, diag.x), "the diagonal-form solve repeats bit for bit"
    ex = _woodbury(C, d, rhs)
    err = float(((diag.x.double() - ex).norm(dim=-2) / ex.norm(dim=-2)).max())
    assert err < 1e-4, err
    if N <= 8192:  # the C oracle's linear_cg with the same preconditioner (the reference's fp32 iteration)
        Lh = host(L)
        xo, _, info = occ.linear_cg(occ.lowrank_diag(C, d), rhs, pre=occ.Preconditioner(Lh, d), n_tridiag=0, tolerance=1e-4)
        assert info.iterations == diag.iterations
        assert max_rel_err_cols(host(diag.x), xo) < 1e-4


def test_diagonal_form_zero_and_nan_right_hand_sides():
    C, d, rhs, desc, pre, _L = _setup(7301, 16, 4096, 32)
    rhs[3] = 0.0
    rd = dev(rhs)
    K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
    res = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
    assert K.cg_last_executed()["rspace_diag"]
    assert not res.nan_detected and (res.x[3] == 0).all()  # linear_cg.py:178-179: a zero column stays zero
    ex = _woodbury(C, d, rhs)
    keep = [i for i in range(16) if i != 3]
    err = float(((res.x.double()[keep] - ex[keep]).norm(dim=-2) / ex[keep].norm(dim=-2)).max())
    assert err < 1e-4
    rhs2 = rhs.copy()
    rhs2[5, 17, 0] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res2 = K.cg_solve(desc, dev(rhs2), precond=pre, tolerance=1e-4)
    assert res2.nan_detected  # :199-200


def test_eigform_is_built_on_the_second_single_column_solve_only():
    C, d, rhs, desc, pre, _L = _setup(7401, 10, 2048, 32)
    full = np.concatenate([rhs, rhs[:, ::-1]], -1)
    K.cg_solve(desc, dev(full), precond=pre, tolerance=1e-4)       # two columns: not counted
    assert pre.rs_uses == 0 and pre.RSD is None
    K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    assert pre.rs_uses == 1 and pre.RSD is None
    K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    assert pre.rs_uses == 2 and torch.is_tensor(pre.RSD)
    os.environ["LO_RS_NO_DIAG"] = "1"
    try:
        K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
        assert not K.cg_last_executed()["rspace_diag"]
    finally:
        del os.environ["LO_RS_NO_DIAG"]
    # the default policy: the form is built when the solves served have paid for it
    K.EIGFORM_AFTER_USES = None
    assert not K._eigform_due(24, 512) and K._eigform_due(25, 512)
    assert not K._eigform_due(100, 40) and K._eigform_due(134, 40)
    assert K._eigform_due(11, 4096) and not K._eigform_due(10, 4096)
    K.EIGFORM_AFTER_USES = -1
    assert not K._eigform_due(10 ** 6, 512)


def test_right_hand_side_inside_span_of_the_root_is_redone_on_the_dense_form():
    """A right-hand side (almost) inside span(C) leaves the complement's coordinate with a weight of tau2 / s in every
    inner product of the diagonal iteration: the kernel flags the member (CgCtrl::rs_redo) and
    lo_cg_solve_f32 repeats the result-only pass on the dense form -- same bits as a dense-form solve."""
    C, d, rhs, desc, pre, _L = _setup(7501, 40, 4096, 32)
    rng = np.random.default_rng(3)
    rhs_in = (C.astype(np.float64) @ rng.standard_normal((40, 32, 1)) + 1e-3 * rhs).astype(np.float32)
    rhs_mixed = rhs.copy()
    rhs_mixed[7] = rhs_in[7]                      # one member of the batch is enough
    dense = K.cg_solve(desc, dev(rhs_mixed), precond=pre, tolerance=1e-4)
    assert not K.cg_last_executed()["rspace_diag"]
    pre.ensure_eigform()
    assert torch.is_tensor(pre.RSD)
    res = K.cg_solve(desc, dev(rhs_mixed), precond=pre, tolerance=1e-4)
    e = K.cg_last_executed()
    assert e["rspace"] == "resident" and not e["rspace_diag"] and e["lean"], e
    assert torch.equal(res.x, dense.x) and res.iterations == dense.iterations
    ex = _woodbury(C, d, rhs_mixed)
    assert float(((res.x.double() - ex).norm(dim=-2) / ex.norm(dim=-2)).max()) < 1e-4
    # the same cache keeps the diagonal form for right-hand sides in general position
    ok = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    assert K.cg_last_executed()["rspace_diag"]
    exo = _woodbury(C, d, rhs)
    assert float(((ok.x.double() - exo).norm(dim=-2) / exo.norm(dim=-2)).max()) < 1e-4


def test_eigform_is_refused_where_the_change_of_basis_would_cost_accuracy():
    """(s_max / s_min)(1 + s_max^2) > 1e9 over E's kept directions: status -2, the cache keeps the dense form."""
    C, d, rhs = cases.lowrank_diag(7601, 6, 4096, 32, 1)
    C[..., 5] *= np.float32(1e-6)
    d = ((d - 0.5) * 0.01 + 0.001).astype(np.float32)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    L, perm = K.pivoted_cholesky(desc, 15)
    pre = K.precond_build(L, dev(d), constant_diag=False, root=desc.A0, perm=perm)
    pre.ensure_eigform()
    assert pre.RSD is None and pre.rsd_refused == "ill-conditioned basis"
    pre.ensure_eigform()                          # (one attempt per cache)
    assert pre.RSD is None and pre.rsd_refused == "ill-conditioned basis"
    K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    e = K.cg_last_executed()
    assert e["rspace"] == "resident" and not e["rspace_diag"]
    ex = _woodbury(C, d, rhs)
    assert float(((res.x.double() - ex).norm(dim=-2) / ex.norm(dim=-2)).max()) < 1e-4


def test_diagonal_form_against_the_real_references_solve_golden_g4():
    """g4_solve_lowrank: the REAL reference's A.solve (linear_cg + rank-15 pivoted-Cholesky preconditioner, its own fp32
    iteration) and the exact solution of the same systems; the diagonal form reproduces the reference's matvec count and
    lands within 1e-4 of both (tests/golden/make_golden.py generated the fixture from /root/reference)."""
    from conftest import load_golden

    g = load_golden("g4_solve_lowrank")
    C, d, rhs = cases.lowrank_diag(401, 4, 2048, 32, 1)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    L, perm = K.pivoted_cholesky(desc, 15, 1e-3)
    pre = K.precond_build(L, dev(d), constant_diag=False, root=desc.A0, perm=perm)
    K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    assert K.cg_last_executed()["rspace_diag"]
    assert res.iterations == int(g["matvecs"]) - 1 == 11 and res.tolerance_reached
    assert max_rel_err_cols(host(res.x), g["x"]) < 1e-4
    assert max_rel_err_cols(host(res.x), g["x_exact"]) < 1e-4
