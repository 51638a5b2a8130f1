"""k_cg_step_cols (csrc/lo_cg_step_cols.hip): the whole step of a streaming CG iteration behind the operator product --
alpha, the residual / solution update, the Woodbury preconditioner (Q form on the matrix cores), beta, the search
direction, the stop rule and the tridiagonal recurrence (reference utils/linear_cg.py:245-332,
operators/added_diag_linear_operator.py:135-140) -- in one launch for up to 32 columns.

  * against the numpy oracle on identical inputs (iterations pinned, solves 1e-4, tridiagonals on the leading block);
  * against the multi-launch streaming iteration (LO_NO_STEP_COLS=1) over group sizes 1 .. 64, one and two column tiles,
    preconditioner ranks 3 / 7 / 15 (row strides 4 / 8 / 16), ragged N, more members than resident groups;
  * the CG coefficients against the float64 run of the same recurrence: as close as the multi-launch path;
  * an injected hand-off timeout: the solve is redone on the multi-launch path and returns its bits.
"""
import numpy as np
import pytest
import torch

import cases
from conftest import max_rel_err_cols

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from oracle import lo_oracle as orc  # noqa: E402  (the checker)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


def _dense_case(seed, B, N, c, k):
    g = torch.Generator(device="cuda").manual_seed(seed)
    R = max(8, min(N // 4, 256))
    X = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    Kd = X @ X.mT
    Kd = ((Kd + Kd.mT) * 0.5).contiguous()
    d = torch.rand(B, N, generator=g, device="cuda") * 0.5 + 0.25
    rhs = torch.randn(B, N, c, generator=g, device="cuda")
    desc = K.dense_diag_descriptor(Kd, d)
    pre = None
    if k:
        L, _ = K.pivoted_cholesky(desc, k)
        pre = K.precond_build(L, d, False)
    return Kd, d, rhs, desc, pre


def _cg64(Kd, d, rhs, pre, iters, nt):
    """linear_cg.py:245-332 in float64 with the same preconditioner: the exact coefficients both fp32 paths approximate"""
    A, dd, b = Kd.double(), d.double(), rhs.double()
    nrm = b.norm(dim=-2, keepdim=True)
    r = b / nrm
    if pre is not None:
        Q, di = pre.Q.double(), pre.dinv.double().unsqueeze(-1)
        prec = lambda v: v * di - Q @ (Q.mT @ v)  # noqa: E731
    else:
        prec = lambda v: v  # noqa: E731
    z = prec(r)
    p = z.clone()
    rz = (r * z).sum(-2, keepdim=True)
    T = torch.zeros(nt, rhs.shape[0], iters, iters, dtype=torch.float64, device=rhs.device)
    pa = pb = None
    for k in range(iters):
        Ap = A @ p + dd.unsqueeze(-1) * p
        al = rz / (p * Ap).sum(-2, keepdim=True)
        r = r - al * Ap
        z = prec(r)
        rzn = (r * z).sum(-2, keepdim=True)
        be = rzn / rz
        rz = rzn
        p = z + be * p
        ar = 1.0 / al[:, 0, :nt]
        if k == 0:
            T[:, :, 0, 0] = ar.T
        else:
            T[:, :, k, k] = (ar + pb * pa).T
            off = (pb.sqrt() * pa).T
            T[:, :, k, k - 1] = off
            T[:, :, k - 1, k] = off
        pa, pb = ar, be[:, 0, :nt]
    return T


@pytest.mark.parametrize("B,N,c,k,nt", [
    (1, 100, 3, 0, 0),       # one workgroup per member, no preconditioner
    (3, 256, 2, 0, 2),
    (2, 700, 5, 3, 4),       # groups of 4, preconditioner rows of 4 floats
    (1, 1000, 11, 7, 10),    # rows of 8 floats
    (2, 4096, 16, 15, 16),   # groups of 16, one full column tile
    (3, 3001, 17, 15, 16),   # ragged N, the second column tile carries one column
    (2, 5000, 32, 10, 8),    # two full column tiles
    (1, 2000, 1, 15, 0),     # single column below the fused apply of lo_precond_fused.hip
    (2, 12000, 11, 15, 10),  # groups of 64
    (150, 520, 4, 5, 3),     # more members than resident groups: the dynamic hand-out
    (1, 20000, 5, 15, 4),    # two row blocks per workgroup (r, p re-read behind the exchange)
    (1, 33001, 2, 0, 0),     # four row blocks, ragged, no preconditioner
])
def test_one_launch_step_equals_the_multi_launch_iteration(monkeypatch, B, N, c, k, nt):
    Kd, d, rhs, desc, pre = _dense_case(1000 + N + c, B, N, c, k)
    res = K.cg_solve(desc, rhs, precond=pre, n_tridiag=nt, tolerance=1e-4)
    plan = K.cg_last_executed()
    assert plan["streaming_precond"] == ("fused_cols" if k else "fused_cols_nopre"), plan
    again = K.cg_solve(desc, rhs, precond=pre, n_tridiag=nt, tolerance=1e-4)
    assert torch.equal(res.x, again.x), "fixed summation order: repeated solves are bit-identical"
    monkeypatch.setenv("LO_NO_STEP_COLS", "1")
    ref = K.cg_solve(desc, rhs, precond=pre, n_tridiag=nt, tolerance=1e-4)
    assert K.cg_last_executed()["streaming_precond"] in ("two_pass", "none")
    monkeypatch.delenv("LO_NO_STEP_COLS")
    assert res.iterations == ref.iterations and res.tolerance_reached == ref.tolerance_reached
    assert max_rel_err_cols(host(res.x), host(ref.x)) < 2e-5
    Ax = Kd @ res.x + d.unsqueeze(-1) * res.x
    assert ((Ax - rhs).norm(dim=-2) / rhs.norm(dim=-2)).max().item() < 5e-4
    if nt and N <= 12000:
        T64 = _cg64(Kd, d, rhs, pre, res.t_mat.shape[-1], nt)
        for lead in (5, 10):
            sc = T64[..., :lead, :lead].abs().amax()
            ea = ((res.t_mat[..., :lead, :lead].double() - T64[..., :lead, :lead]).abs().amax() / sc).item()
            eb = ((ref.t_mat[..., :lead, :lead].double() - T64[..., :lead, :lead]).abs().amax() / sc).item()
            assert ea <= max(3 * eb, 2e-6), (lead, ea, eb)


@pytest.mark.parametrize("c,nt,k", [(11, 10, 15), (17, 16, 15), (4, 0, 0)])
def test_one_launch_step_against_the_oracle(c, nt, k):
    """One dense operator of 1500 rows (a single GP of GPyTorch's default probe count at c = 11) against the numpy
    restatement of linear_cg on identical inputs: iteration count, solves, tridiagonals, pivots of the preconditioner."""
    N, B = 1500, 2
    M = np.stack([cases.spd_test_matrix(300 + i, N, dtype=np.float32) for i in range(B)])
    M = (M / np.abs(M).max()).astype(np.float32)
    dgn = (cases.randn(310, B, N, dtype=np.float32) ** 2 * 0.1 + 0.05).astype(np.float32)
    rhs = cases.randn(311, B, N, c, dtype=np.float32)
    desc = K.dense_diag_descriptor(dev(M), dev(dgn))
    pre, pre_o = None, None
    if k:
        L, piv = K.pivoted_cholesky(desc, k)
        Lo, pivo = orc.pivoted_cholesky(orc.DenseRowSource(M), k)
        assert np.array_equal(host(piv)[..., :k], pivo[..., :k])
        pre = K.precond_build(L, dev(dgn), False)
        pre_o = orc.Preconditioner(Lo, dgn)
    res = K.cg_solve(desc, dev(rhs), precond=pre, n_tridiag=nt, tolerance=1e-4)
    assert K.cg_last_executed()["streaming_precond"] == ("fused_cols" if k else "fused_cols_nopre")
    xo, to, info = orc.linear_cg(lambda v: orc.matvec_dense_diag(M, dgn, v), rhs, n_tridiag=nt, tolerance=1e-4,
                                 preconditioner=(pre_o.apply if pre_o is not None else None))
    assert res.iterations == info.iterations
    assert max_rel_err_cols(host(res.x), xo) < 1e-4
    if nt:
        t, t_o = host(res.t_mat).astype(np.float64), np.asarray(to, dtype=np.float64)
        lead = 6
        blk = t_o[..., :lead, :lead]
        assert (np.abs(t[..., :lead, :lead] - blk) / (np.abs(blk) + 1e-2 * np.abs(blk).max())).max() < 1e-3


@pytest.mark.parametrize("B,n,c", [(3, 64, 5), (2, 200, 3)])
def test_kronecker_columns_take_the_one_launch_step(B, n, c):
    """Kronecker operator with a constant diagonal, several columns: the step kernel does not depend on the operator
    (200 (x) 200: 40000 rows, four row blocks per workgroup)."""
    g = torch.Generator(device="cuda").manual_seed(77)
    X1 = torch.randn(B, n, n, generator=g, device="cuda") / n ** 0.5
    X2 = torch.randn(B, n, n, generator=g, device="cuda") / n ** 0.5
    K1 = X1 @ X1.mT + 0.1 * torch.eye(n, device="cuda")
    K2 = X2 @ X2.mT + 0.1 * torch.eye(n, device="cuda")
    sig = torch.full((B,), 0.05, device="cuda")
    rhs = torch.randn(B, n * n, c, generator=g, device="cuda")
    desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
    L, _ = K.pivoted_cholesky(desc.without_diag(), 15)
    pre = K.precond_build(L, sig, True)
    res = K.cg_solve(desc, rhs, precond=pre, n_tridiag=2, tolerance=1e-4)
    assert K.cg_last_executed()["streaming_precond"] == "fused_cols"
    Ax = torch.einsum("bij,bjkc->bikc", K1, torch.einsum("bkl,bjlc->bjkc", K2, res.x.reshape(B, n, n, c))).reshape(B, n * n, c)
    Ax = Ax + 0.05 * res.x
    assert ((Ax - rhs).norm(dim=-2) / rhs.norm(dim=-2)).max().item() < 1e-3
    assert res.tolerance_reached


def test_injected_timeout_redoes_the_solve_on_the_multi_launch_path(monkeypatch, capfd):
    Kd, d, rhs, desc, pre = _dense_case(4242, 2, 1500, 6, 7)
    monkeypatch.setenv("LO_NO_STEP_COLS", "1")
    ref = K.cg_solve(desc, rhs, precond=pre, n_tridiag=4, tolerance=1e-4)
    monkeypatch.delenv("LO_NO_STEP_COLS")
    monkeypatch.setenv("LO_SC_TEST_FALLBACK", "2")
    res = K.cg_solve(desc, rhs, precond=pre, n_tridiag=4, tolerance=1e-4)
    monkeypatch.delenv("LO_SC_TEST_FALLBACK")
    assert "redoing the solve" in capfd.readouterr().err
    assert K.cg_last_executed()["streaming_precond"] == "two_pass"
    assert res.iterations == ref.iterations and torch.equal(res.x, ref.x) and torch.equal(res.t_mat, ref.t_mat)
