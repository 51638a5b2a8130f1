"""Pins the C oracle (oracle/lo_oracle_c.c -- the CPU restatement with the argument structures of include/lo_amd.h,
"a CPU build of the same ABI" kept as test infrastructure) against the golden vectors the REAL reference produced and
against the numpy oracle: pivots / permutations and the factor L bit for bit, floats within the tolerance at each assert."""
import numpy as np
import pytest

import cases
from conftest import load_golden, max_rel_err_cols, rel_err, tridiag_block_err
from oracle import lo_oracle as orc
from oracle import lo_oracle_c as occ


def test_library_builds_and_reports_threads():
    assert occ.num_threads() >= 1


def test_matvecs_against_golden_g6_and_numpy():
    C, d, v = cases.lowrank_diag(601, (2, 3), 200, 8, 4)
    y = occ.matvec(occ.lowrank_diag(C, d.reshape(6, 200)), v.reshape(6, 200, 4))
    assert max_rel_err_cols(y, orc.matvec_lowrank_diag(C, d, v).reshape(6, 200, 4)) < 2e-6
    K, dd, vv = cases.dense_diag(602, 3, 150, 5)
    assert max_rel_err_cols(occ.matvec(occ.dense_diag(K, dd), vv), orc.matvec_dense_diag(K, dd, vv)) < 2e-6
    K1, K2, sig, vk = cases.kron_factors(603, 3, 12, 10, 3)
    yk = occ.matvec(occ.kron_diag(K1, K2, sig[:, 0], const_diag=True), vk)
    dk = np.broadcast_to(sig, (3, 120)).astype(np.float32)
    assert max_rel_err_cols(yk, orc.matvec_kron_diag(K1, K2, dk, vk)) < 2e-6
    # a sum of two structured terms + one diagonal (sum_linear_operator.py:47-51)
    C2 = cases.lowrank_diag(604, 3, 150, 4, 1)[0]
    ys = occ.matvec(occ.sum_op([occ.lowrank_diag(C2), occ.dense_diag(K)], d=dd), vv)
    ref = orc.matvec_lowrank_diag(C2, None, vv) + orc.matvec_dense_diag(K, None, vv) + dd[..., None] * vv
    assert max_rel_err_cols(ys, ref) < 2e-6


def test_pivoted_cholesky_bit_exact_against_goldens_and_numpy():
    g = load_golden("g2_pivchol_lowrank")
    for R in (8, 32):
        C, _, _ = cases.lowrank_diag(210 + R, 3, 2048, R, 1)
        L, perm = occ.pivoted_cholesky(occ.lowrank_diag(C), 15)
        Lo, po = orc.pivoted_cholesky(orc.LowRankRowSource(C), 15)
        assert np.array_equal(perm, po) and np.array_equal(L, Lo), R   # same operation order: the same bits
        assert np.array_equal(perm, g[f"piv_R{R}"]) and L.shape == g[f"L_R{R}"].shape   # the real reference's pivots
        assert np.allclose(L, g[f"L_R{R}"], rtol=1e-4, atol=1e-5)
    # dense and Kronecker row sources, and a sum of two
    K, _, _ = cases.dense_diag(221, 2, 300, 1)
    L, perm = occ.pivoted_cholesky(occ.dense_diag(K), 10)
    Lo, po = orc.pivoted_cholesky(orc.DenseRowSource(K), 10)
    assert np.array_equal(perm, po) and np.array_equal(L, Lo)
    K1, K2, _, _ = cases.kron_factors(222, 2, 16, 16, 1)
    L, perm = occ.pivoted_cholesky(occ.kron_diag(K1, K2), 15)
    Lo, po = orc.pivoted_cholesky(orc.KronRowSource(K1, K2), 15)
    assert np.array_equal(perm, po) and np.array_equal(L, Lo)
    C2 = cases.lowrank_diag(223, 2, 300, 6, 1)[0]
    L, perm = occ.pivoted_cholesky(occ.sum_op([occ.lowrank_diag(C2), occ.dense_diag(K)]), 12)
    Lo, po = orc.pivoted_cholesky(orc.SumRowSource(orc.LowRankRowSource(C2), orc.DenseRowSource(K)), 12)
    assert np.array_equal(perm, po) and np.array_equal(L, Lo)


def test_pivoted_cholesky_reference_recipe_dense8():
    """test/functions/test_pivoted_cholesky.py:24-62 recipes (8 x 8 and [2, 3, 8, 8] dense, seed 0) from the real reference."""
    g = load_golden("g2_pivchol_dense8")
    m8, mb = cases.pivchol_dense8(201), cases.pivchol_dense8(202, batch=(2, 3))
    L, piv = occ.pivoted_cholesky(occ.dense_diag(m8[None]), 3)
    assert np.array_equal(piv[0], g["piv"]) and np.allclose(L[0], g["L"], rtol=1e-5, atol=1e-6)
    Lb, pivb = occ.pivoted_cholesky(occ.dense_diag(mb.reshape(6, 8, 8)), 3)
    assert np.array_equal(pivb.reshape(2, 3, 8), g["pivb"]) and np.allclose(Lb.reshape(2, 3, 8, 3), g["Lb"], rtol=1e-5, atol=1e-6)
    L8, piv8 = occ.pivoted_cholesky(occ.dense_diag(m8[None]), 8)
    assert np.array_equal(piv8[0], g["piv8"]) and np.allclose(L8[0], g["L8"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("const", [False, True])
def test_preconditioner_matches_the_qr_form(const):
    C, d, rhs = cases.lowrank_diag(301, 3, 2048, 32, 4)
    if const:
        d = np.broadcast_to(d[:, :1], d.shape).copy()
    L, _ = orc.pivoted_cholesky(orc.LowRankRowSource(C), 15)
    po = orc.Preconditioner(L, d)
    pc = occ.Preconditioner(L, d[:, 0] if const else d, const_diag=const)
    assert po.constant_diag == const
    assert np.allclose(pc.logdet, po.logdet, rtol=1e-5)
    assert max_rel_err_cols(pc.apply(rhs), po.apply(rhs)) < 1e-5


def test_linear_cg_against_golden_g1_and_g4():
    g = load_golden("g1_cg_fp32_lowrank")
    C, d, rhs = cases.lowrank_diag(141, 4, 512, 8, 5)
    op = occ.lowrank_diag(C, d)
    x, _, info = occ.linear_cg(op, rhs, tolerance=1.0)
    assert info.matvecs == int(g["matvecs_tol1"]) == 12 and info.iterations == 11 and info.tolerance_reached
    # (unpreconditioned fp32 CG at the floor carries ~2e-5 of summation-order noise between any two fp32 implementations
    #  -- numpy vs ATen just the same: tests/test_oracle_vs_golden.py; the bar is north_star's 1e-4)
    assert max_rel_err_cols(x, g["x_tol1"]) < 1e-4
    x, t, info = occ.linear_cg(op, rhs, tolerance=1.0, n_tridiag=4)
    assert info.matvecs == int(g["matvecs_t_tol1"]) == 22 and t.shape == g["t_mat_tol1"].shape
    assert max_rel_err_cols(x, g["xt_tol1"]) < 1e-4
    g23 = load_golden("g23_tridiag_divergence_tight_logdet")
    err, k = tridiag_block_err(t, g23["g1_t_mat_f64"], g23["g1_valid"], back_off=1)
    assert k >= 9 and err < 3e-4, (err, k)
    # zero column + initial guess
    rhs_z = rhs.copy()
    rhs_z[1, :, 2] = 0.0
    x0 = cases.randn(142, 4, 512, 5, dtype=np.float32) * 0.1
    xz, _, iz = occ.linear_cg(op, rhs_z, x0=x0, tolerance=1e-4)
    assert abs(iz.matvecs - int(g["matvecs_zero_col"])) <= 1
    assert max_rel_err_cols(np.delete(xz, 2, -1), np.delete(g["x_zero_col"], 2, -1)) < 1e-4
    # cfg2-shaped solve with the default preconditioner (golden from A.solve of the real reference)
    g4 = load_golden("g4_solve_lowrank")
    C, d, rhs = cases.lowrank_diag(401, 4, 2048, 32, 1)
    op = occ.lowrank_diag(C, d)
    L, _ = occ.pivoted_cholesky(occ.lowrank_diag(C), 15)
    x, _, info = occ.linear_cg(op, rhs, pre=occ.Preconditioner(L, d), tolerance=1e-4)
    assert info.matvecs == int(g4["matvecs"]) and max_rel_err_cols(x, g4["x"]) < 1e-5
    assert max_rel_err_cols(x, g4["x_exact"]) < 1e-5


def test_inv_quad_logdet_pipeline_against_golden_g4_and_numpy():
    g = load_golden("g4_iql_lowrank")
    C, d, rhs = cases.lowrank_diag(411, 3, 2048, 16, 1)
    Z, _ = cases.probes(412, 3, 2048, 8)
    L, perm = occ.pivoted_cholesky(occ.lowrank_diag(C), 15)
    pre = occ.Preconditioner(L, d)
    x, t, info = occ.linear_cg(occ.lowrank_diag(C, d), np.concatenate([Z, rhs], -1), pre=pre, n_tridiag=8, tolerance=1e-4)
    assert info.matvecs == int(g["matvecs"]) == 22
    assert max_rel_err_cols(x, g["solves"]) < 1e-5
    assert np.allclose(pre.logdet, g["logdet_p"], rtol=1e-5)
    assert np.allclose((x[..., 8:] * rhs).sum(-2)[..., 0], g["inv_quad"], rtol=1e-5)
    assert rel_err(t[..., :2, :2], g["t_mat"][..., :2, :2]) < 1e-4
    evals, evecs = orc.lanczos_tridiag_to_diag(t.astype(np.float64))
    assert np.allclose(orc.slq_logdet(2048, evals, evecs) + pre.logdet, g["logdet"], rtol=1e-4, atol=2048 * 1.2e-7 * 137.0)
    # a Kronecker solve that runs far beyond the floor: iteration count within the reference's +-3
    gk = load_golden("g4_solve_kron")
    K1, K2, sig, rk = cases.kron_factors(421, 2, 48, 48, 1)
    Lk, _ = occ.pivoted_cholesky(occ.kron_diag(K1, K2), 15)
    pk = occ.Preconditioner(Lk, sig[:, 0], const_diag=True)
    xk, _, ik = occ.linear_cg(occ.kron_diag(K1, K2, sig[:, 0], const_diag=True), rk, pre=pk, tolerance=1e-3)
    assert abs(ik.matvecs - int(gk["matvecs"])) <= 3 and max_rel_err_cols(xk, gk["x"]) < 5e-3


def test_lanczos_against_golden_g5_and_numpy():
    """lo_cpu_lanczos_tridiag_f32 (utils/lanczos.py:9-164) against the real reference's golden g5 and the numpy oracle:
    same shapes (the batch-global early exit included), tridiagonals to 1e-5 of the numpy restatement (same arithmetic,
    other summation order), the reference's own acceptance Q T Q^T = M (test/utils/test_lanczos.py:35-36)."""
    g = load_golden("g5_lanczos")
    M = cases.spd_test_matrix(501, 100, dtype=np.float32, jitter=1e-6)
    v0 = cases.randn(502, 100, 1, dtype=np.float32)
    q, t = occ.lanczos_tridiag(occ.dense_diag(M[None]), v0[None], 100)
    q, t = q[0], t[0]  # (one member)
    assert q.shape == g["q_near"].shape and t.shape == g["t_near"].shape
    assert np.allclose(t[:10, :10], g["t_near"][:10, :10], rtol=1e-3, atol=1e-5)
    assert np.allclose(q @ t @ q.T, M, atol=1e-4)
    C, d, _ = cases.lowrank_diag(511, 2, 256, 8, 1)
    V = cases.randn(512, 2, 256, 3, dtype=np.float32)
    q3, t3 = occ.lanczos_tridiag(occ.lowrank_diag(C, d), V, 10)
    qo, to = orc.lanczos_tridiag(lambda v: orc.matvec_lowrank_diag(C, d, v), 10, V)
    assert q3.shape == g["q_batch"].shape == qo.shape and t3.shape == g["t_batch"].shape == to.shape
    assert np.allclose(t3, to, rtol=1e-4, atol=1e-5) and np.allclose(q3, qo, atol=2e-4)
    assert np.allclose(t3, g["t_batch"], rtol=1e-3, atol=1e-4) and np.allclose(q3, g["q_batch"], atol=2e-3)
    # early exit: an operator of rank 3 + identity exhausts its Krylov space after four steps in every member
    C4 = cases.lowrank_diag(513, 2, 200, 3, 1)[0]
    ones = np.ones((2, 200), dtype=np.float32)
    V4 = cases.randn(514, 2, 200, 2, dtype=np.float32)
    q4, t4 = occ.lanczos_tridiag(occ.lowrank_diag(C4, ones), V4, 20)
    qo4, to4 = orc.lanczos_tridiag(lambda v: orc.matvec_lowrank_diag(C4, ones, v), 20, V4)
    assert t4.shape == to4.shape and t4.shape[-1] < 20


def test_tridiag_eigh_slq_against_numpy_and_golden_g4():
    """lo_cpu_tridiag_eigh_slq_f32 (implicit QL in double; utils/lanczos.py:167-189 + stochastic_lq.py:45-82): eigenvalues
    and first eigenvector components against numpy's eigh in float64 on random tridiagonals (clamping of negative
    eigenvalues included), and the SLQ value of the REFERENCE's own t_mat (golden g4) against the reference's number --
    within the fp32 eigensolver's noise floor the reference's value carries (test_oracle_vs_golden.py: `floor`)."""
    rng = np.random.default_rng(77)
    P, B, k = 3, 4, 12
    t = np.zeros((P, B, k, k), dtype=np.float32)
    dg = rng.uniform(0.5, 3.0, (P, B, k)).astype(np.float32)
    off = rng.uniform(-1.0, 1.0, (P, B, k - 1)).astype(np.float32)
    dg[0, 0, 3] = -2.0  # an indefinite block: negative eigenvalues are clamped
    for i in range(k):
        t[..., i, i] = dg[..., i]
    for i in range(k - 1):
        t[..., i, i + 1] = t[..., i + 1, i] = off[..., i]
    ld, ev, v0 = occ.tridiag_eigh_slq(t, 500, want_spectrum=True)
    w, v = np.linalg.eigh(t.astype(np.float64))
    mask = w >= 0
    first = v[..., 0, :] * mask
    wc = np.where(mask, w, 1.0)
    order = np.argsort(ev, axis=-1)
    ev_s, v0_s = np.take_along_axis(ev, order, -1), np.take_along_axis(v0, order, -1)
    wo = np.argsort(wc, axis=-1)
    assert np.allclose(ev_s, np.take_along_axis(wc, wo, -1), rtol=1e-12, atol=1e-12)
    want = (500.0 / P) * np.sum(first ** 2 * np.log(wc), axis=(0, -1))
    assert np.allclose(ld, want, rtol=1e-6)
    assert np.allclose(np.sort(v0_s ** 2, -1), np.sort(np.take_along_axis(first, wo, -1) ** 2, -1), atol=1e-10)
    g = load_golden("g4_iql_lowrank")
    got = occ.tridiag_eigh_slq(g["t_mat"], 2048)
    floor = 2048 * 1.2e-7 * 137.0
    assert np.allclose(got, g["pinvk_logdet"], rtol=1e-4, atol=floor)
    # ... and EXACTLY (1e-6) the float64 eigendecomposition of the same matrices
    ev64, evec64 = orc.lanczos_tridiag_to_diag(g["t_mat"].astype(np.float64))
    assert np.allclose(got, orc.slq_logdet(2048, ev64, evec64), rtol=2e-6)


def test_inv_quad_logdet_in_c_against_golden_g4_and_numpy():
    """The whole forward of InvQuadLogdet with injected probes in C (pivoted Cholesky -> preconditioner -> linear_cg with
    tridiagonals -> eigh + SLQ) against the real reference's golden g4_iql_lowrank and the numpy oracle."""
    g = load_golden("g4_iql_lowrank")
    C, d, rhs = cases.lowrank_diag(411, 3, 2048, 16, 1)
    Z, _ = cases.probes(412, 3, 2048, 8)
    iq, ld, solves, t_mat, info, piv = occ.inv_quad_logdet(occ.lowrank_diag(C, d), occ.lowrank_diag(C), d, rhs, Z, tolerance=1e-4)
    assert info.matvecs == int(g["matvecs"])
    assert max_rel_err_cols(solves, g["solves"]) < 1e-5
    assert np.allclose(iq[..., 0], g["inv_quad"], rtol=1e-5)
    floor = 2048 * 1.2e-7 * 137.0
    assert np.allclose(ld, g["logdet"], rtol=1e-4, atol=floor)
    iqo, ldo, so, to, infoo, _ = orc.inv_quad_logdet(lambda v: orc.matvec_lowrank_diag(C, d, v), orc.LowRankRowSource(C), d, rhs, Z,
                                                     tolerance=1e-4)
    assert infoo.matvecs == info.matvecs and np.allclose(iq, iqo, rtol=1e-5)
    assert np.allclose(ld, ldo, rtol=1e-4, atol=floor)
