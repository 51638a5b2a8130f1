#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz by running the REAL reference.

Runs only in the build container (imports cornellius-gp/linear_operator from /root/reference;
nothing of the reference travels to the GPU box -- only the .npz outputs committed next to this
script).  Usage:  python tests/golden/make_golden.py

Inputs come from tests/golden/cases.py (numpy PCG64, seeded); each file stores the reference's
outputs, the iteration / matvec counts observed through a spy on `linear_operator.utils.linear_cg`
(the seam the reference's own tests patch, linear_operator/test/linear_operator_test_case.py:555),
and a checksum of the inputs.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import torch  # noqa: E402

import linear_operator  # noqa: E402
from linear_operator import settings  # noqa: E402
from linear_operator.operators import (  # noqa: E402
    AddedDiagLinearOperator,
    ConstantDiagLinearOperator,
    DenseLinearOperator,
    DiagLinearOperator,
    KroneckerProductLinearOperator,
    LowRankRootLinearOperator,
    PsdSumLinearOperator,
    SumLinearOperator,
)
from linear_operator.utils import linear_cg as _ref_linear_cg  # noqa: E402
from linear_operator.utils.lanczos import lanczos_tridiag, lanczos_tridiag_to_diag  # noqa: E402
from linear_operator.utils.stochastic_lq import StochasticLQ  # noqa: E402
from linear_operator.utils.warnings import NumericalWarning  # noqa: E402

import cases  # noqa: E402

torch.set_num_threads(8)
T = torch.from_numpy


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


class Counter:
    def __init__(self, fn):
        self.fn = fn
        self.calls = 0

    def __call__(self, x):
        self.calls += 1
        return self.fn(x)


def run_cg(closure, rhs, **kw):
    cnt = Counter(closure)
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        out = _ref_linear_cg(cnt, rhs, **kw)
    warned = any(issubclass(w.category, NumericalWarning) for w in ws)
    return out, cnt.calls, warned


# ------------------------------------------------------------------------------------------------
def g1_linear_cg():
    print("G1 linear_cg")
    # test_linear_cg.py:27-64 recipe, fp64 N=100
    M = cases.spd_test_matrix(101, 100)
    b_vec = cases.randn(102, 100)
    b_mat = cases.randn(103, 100, 50)
    x0_vec = cases.randn(104, 100)
    x0_mat = cases.randn(105, 100, 50)
    Mt = T(M)
    (x_vec), n1, w1 = run_cg(Mt.matmul, T(b_vec), max_iter=100)
    (x_vec_i), n2, w2 = run_cg(Mt.matmul, T(b_vec), max_iter=100, initial_guess=T(x0_vec))
    (x_mat), n3, w3 = run_cg(Mt.matmul, T(b_mat), max_iter=100)
    (x_mat_i), n4, w4 = run_cg(Mt.matmul, T(b_mat), max_iter=100, initial_guess=T(x0_mat))
    save("g1_cg_fp64_n100", x_vec=x_vec, x_vec_init=x_vec_i, x_mat=x_mat, x_mat_init=x_mat_i,
         matvecs=np.array([n1, n2, n3, n4]), warned=np.array([w1, w2, w3, w4]),
         checksum=cases.checksum(M, b_vec, b_mat, x0_vec, x0_mat))

    # test_linear_cg.py:66-95 recipe, fp64 N=10, c=50, n_tridiag=5
    M = cases.spd_test_matrix(111, 10)
    b = cases.randn(112, 10, 50)
    (x, t), n, w = run_cg(T(M).matmul, T(b), n_tridiag=5, max_tridiag_iter=10, max_iter=10, tolerance=0, eps=1e-15)
    save("g1_cg_fp64_n10_tridiag", x=x, t_mat=t, matvecs=n, warned=w, checksum=cases.checksum(M, b))

    # test_linear_cg.py:97-144 recipes, batch 5
    M = cases.spd_test_matrix(121, 100, batch=(5,))
    b = cases.randn(122, 5, 100, 50)
    (x), n, w = run_cg(T(M).matmul, T(b), max_iter=100)
    save("g1_cg_fp64_batch", x=x, matvecs=n, warned=w, checksum=cases.checksum(M, b))
    M = cases.spd_test_matrix(131, 10, batch=(5,))
    b = cases.randn(132, 5, 10, 10)
    (x, t), n, w = run_cg(T(M).matmul, T(b), n_tridiag=8, max_iter=10, max_tridiag_iter=10, tolerance=0, eps=1e-30)
    save("g1_cg_fp64_batch_tridiag", x=x, t_mat=t, matvecs=n, warned=w, checksum=cases.checksum(M, b))

    # fp32 low-rank + diag, B=4, N=512, R=8, c=5, no preconditioner, n_tridiag=4, tolerances {1, 1e-4}
    C, d, rhs = cases.lowrank_diag(141, 4, 512, 8, 5)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    out = {}
    for tag, tol in (("tol1", 1.0), ("tol1e4", 1e-4)):
        (x), n, w = run_cg(A._matmul, T(rhs), tolerance=tol)
        (x2, t2), n2, w2 = run_cg(A._matmul, T(rhs), tolerance=tol, n_tridiag=4)
        out.update({f"x_{tag}": x, f"matvecs_{tag}": n, f"warned_{tag}": w,
                    f"xt_{tag}": x2, f"t_mat_{tag}": t2, f"matvecs_t_{tag}": n2, f"warned_t_{tag}": w2})
    # zero column + initial guess
    rhs_z = rhs.copy()
    rhs_z[1, :, 2] = 0.0
    x0 = cases.randn(142, 4, 512, 5, dtype=np.float32) * 0.1
    (xz), nz, wz = run_cg(A._matmul, T(rhs_z), tolerance=1e-4, initial_guess=T(x0))
    out.update(x_zero_col=xz, matvecs_zero_col=nz)
    save("g1_cg_fp32_lowrank", checksum=cases.checksum(C, d, rhs), **out)


# ------------------------------------------------------------------------------------------------
def g2_pivoted_cholesky():
    print("G2 pivoted_cholesky")
    m8 = cases.pivchol_dense8(201)
    L, piv = linear_operator.pivoted_cholesky(T(m8), rank=3, return_pivots=True)
    mb = cases.pivchol_dense8(202, batch=(2, 3))
    Lb, pivb = linear_operator.pivoted_cholesky(T(mb), rank=3, return_pivots=True)
    L8, piv8 = linear_operator.pivoted_cholesky(T(m8), rank=8, return_pivots=True)
    save("g2_pivchol_dense8", L=L, piv=piv, Lb=Lb, pivb=pivb, L8=L8, piv8=piv8, checksum=cases.checksum(m8, mb))

    out = {}
    cs = []
    for R in (8, 32):
        C, d, _ = cases.lowrank_diag(210 + R, 3, 2048, R, 1)
        op = LowRankRootLinearOperator(T(C))
        Lr, pr = op.pivoted_cholesky(rank=15, return_pivots=True)
        out[f"L_R{R}"] = Lr
        out[f"piv_R{R}"] = pr
        cs += [C]
    save("g2_pivchol_lowrank", checksum=cases.checksum(*cs), **out)

    K1, K2, _, _ = cases.kron_factors(221, 2, 16, 16, 1)
    op = KroneckerProductLinearOperator(T(K1), T(K2))
    Lk, pk = op.pivoted_cholesky(rank=15, return_pivots=True)
    Kd, _, _ = cases.dense_diag(222, 2, 300, 1)
    Ld, pd_ = DenseLinearOperator(T(Kd)).pivoted_cholesky(rank=15, return_pivots=True)
    save("g2_pivchol_kron_dense", L_kron=Lk, piv_kron=pk, L_dense=Ld, piv_dense=pd_,
         checksum=cases.checksum(K1, K2, Kd))


# ------------------------------------------------------------------------------------------------
def g3_preconditioner():
    print("G3 preconditioner")
    out = {}
    C, d, rhs = cases.lowrank_diag(301, 3, 2048, 32, 4)
    with settings.min_preconditioning_size(0):
        # non-constant diag
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
        fn, plt, logdet_p = A._preconditioner()
        out.update(z_nonconst=fn(T(rhs)), logdet_nonconst=logdet_p, L_nonconst=A._piv_chol_self,
                   const_flag_nonconst=A._constant_diag)
        # constant diag
        sig = np.array([[0.3], [0.7], [1.1]], dtype=np.float32)
        A2 = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), ConstantDiagLinearOperator(T(sig), 2048))
        fn2, plt2, logdet_p2 = A2._preconditioner()
        out.update(z_const=fn2(T(rhs)), logdet_const=logdet_p2, const_flag_const=A2._constant_diag,
                   logdet_dense_const=np.linalg.slogdet(plt2.to_dense().numpy().astype(np.float64))[1].astype(np.float32))
    save("g3_precond", checksum=cases.checksum(C, d, rhs), **out)


# ------------------------------------------------------------------------------------------------
class _Spy:
    """Wraps linear_operator.utils.linear_cg, recording matvec count and raw outputs."""

    def __init__(self):
        self.records = []

    def __call__(self, matmul_closure, rhs, **kw):
        cnt = Counter(matmul_closure)
        out = _ref_linear_cg(cnt, rhs, **kw)
        self.records.append(dict(matvecs=cnt.calls, out=out, kw=kw))
        return out


class _ProbedAddedDiag(AddedDiagLinearOperator):
    _probes = None

    def _probe_vectors_and_norms(self):  # hook: operators/_linear_operator.py:629-633
        return self._probes


def _with_spy(fn):
    spy = _Spy()
    old = linear_operator.utils.linear_cg
    linear_operator.utils.linear_cg = spy
    try:
        with warnings.catch_warnings(record=True) as ws:
            warnings.simplefilter("always")
            res = fn()
    finally:
        linear_operator.utils.linear_cg = old
    warned = any(issubclass(w.category, NumericalWarning) for w in ws)
    return res, spy, warned


def g4_solve_and_inv_quad_logdet():
    print("G4 solve / inv_quad_logdet (operator API, default preconditioner)")
    # ---- cfg2-shaped: AddedDiag(LowRankRoot, Diag).solve ----
    C, d, rhs = cases.lowrank_diag(401, 4, 2048, 32, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    with settings.cg_tolerance(1e-4):
        x, spy, w = _with_spy(lambda: A.solve(T(rhs)))
    woodbury = (LowRankRootLinearOperator(T(C).double()) + DiagLinearOperator(T(d).double())).solve(T(rhs).double())
    save("g4_solve_lowrank", x=x, matvecs=spy.records[0]["matvecs"], warned=w, x_exact=woodbury.float(),
         checksum=cases.checksum(C, d, rhs))

    # ---- cfg3-shaped: inv_quad_logdet with injected probes ----
    C, d, rhs = cases.lowrank_diag(411, 3, 2048, 16, 1)
    Z, Zn = cases.probes(412, 3, 2048, 8)
    A = _ProbedAddedDiag(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    A._probes = (T(Z), T(Zn))
    with settings.cg_tolerance(1e-4):
        (iq, ld), spy, w = _with_spy(lambda: A.inv_quad_logdet(T(rhs), logdet=True))
    solves, t_mat = spy.records[0]["out"]
    evals, evecs = lanczos_tridiag_to_diag(t_mat)
    (pinvk_logdet,) = StochasticLQ().to_dense(A.matrix_shape, evals, evecs, [lambda x: x.log()])
    _, _, logdet_p = A._preconditioner()
    dense = (T(C).double() @ T(C).double().mT) + torch.diag_embed(T(d).double())
    save("g4_iql_lowrank", inv_quad=iq, logdet=ld, solves=solves, t_mat=t_mat, evals=evals,
         pinvk_logdet=pinvk_logdet, logdet_p=logdet_p, matvecs=spy.records[0]["matvecs"], warned=w,
         logdet_exact=np.linalg.slogdet(dense.numpy())[1].astype(np.float32),
         checksum=cases.checksum(C, d, rhs, Z))

    # ---- cfg4-shaped: AddedDiag(Kron, ConstantDiag).solve, N = 48*48 = 2304 ----
    K1, K2, sig, rhs = cases.kron_factors(421, 2, 48, 48, 1)
    A = AddedDiagLinearOperator(KroneckerProductLinearOperator(T(K1), T(K2)), ConstantDiagLinearOperator(T(sig), 2304))
    with settings.cg_tolerance(1e-3):
        x, spy, w = _with_spy(lambda: A.solve(T(rhs)))
    dense = torch.stack([torch.kron(T(K1)[i].double(), T(K2)[i].double()) for i in range(2)])
    dense = dense + sig.astype(np.float64)[0, 0] * torch.eye(2304, dtype=torch.float64)
    save("g4_solve_kron", x=x, matvecs=spy.records[0]["matvecs"], warned=w,
         x_exact=np.linalg.solve(dense.numpy(), rhs.astype(np.float64)).astype(np.float32), checksum=cases.checksum(K1, K2, sig, rhs))

    # ---- cfg5-shaped: Dense.add_diagonal(d) -> AddedDiag; inv_quad_logdet with probes, N = 2048 ----
    K, d, rhs = cases.dense_diag(431, 2, 2048, 1)
    Z, Zn = cases.probes(432, 2, 2048, 4)
    base = DenseLinearOperator(T(K)).add_diagonal(T(d))
    assert type(base) is AddedDiagLinearOperator
    A = _ProbedAddedDiag(DenseLinearOperator(T(K)), DiagLinearOperator(T(d)))
    A._probes = (T(Z), T(Zn))
    with settings.cg_tolerance(1e-4):
        (iq, ld), spy, w = _with_spy(lambda: A.inv_quad_logdet(T(rhs), logdet=True))
    solves, t_mat = spy.records[0]["out"]
    save("g4_iql_dense", inv_quad=iq, logdet=ld, solves=solves, t_mat=t_mat, matvecs=spy.records[0]["matvecs"],
         warned=w, checksum=cases.checksum(K, d, rhs, Z))

    # ---- cfg1: Dense 256x256, torch.linalg.solve -> Cholesky branch (functions/_solve.py:17-18) ----
    M = cases.spd_test_matrix(441, 256, dtype=np.float32, jitter=1.0)
    b = cases.randn(442, 256, 3, dtype=np.float32)
    x, spy, w = _with_spy(lambda: torch.linalg.solve(DenseLinearOperator(T(M)), T(b)))
    assert len(spy.records) == 0  # CG must NOT run
    save("g4_cfg1_dense256", x=x, checksum=cases.checksum(M, b))


# ------------------------------------------------------------------------------------------------
def g5_lanczos():
    print("G5 lanczos_tridiag")
    out = {}
    # test_lanczos.py:42-48 recipe (near exact), fp32 N=100, supplied init vec
    M = cases.spd_test_matrix(501, 100, dtype=np.float32, jitter=1e-6)
    v0 = cases.randn(502, 100, 1, dtype=np.float32)
    q, t = lanczos_tridiag(T(M).matmul, max_iter=100, dtype=torch.float32, device=torch.device("cpu"),
                           matrix_shape=M.shape, init_vecs=T(v0))
    out.update(q_near=q, t_near=t)
    # test_lanczos.py:50-57 recipe: orthogonal * diag(10^-i) * orthogonal^T, N=30
    from scipy.stats import ortho_group

    O = ortho_group.rvs(30, random_state=503).astype(np.float32)
    Dg = np.diag(np.array([10.0 ** -i for i in range(30)], dtype=np.float32))
    M2 = (O @ (Dg @ O.T)).astype(np.float32)
    v2 = cases.randn(504, 30, 1, dtype=np.float32)
    q2, t2 = lanczos_tridiag(T(M2).matmul, max_iter=30, dtype=torch.float32, device=torch.device("cpu"),
                             matrix_shape=M2.shape, init_vecs=T(v2))
    out.update(q_approx=q2, t_approx=t2, M_approx=M2)
    # batched low-rank + diag, B=2, N=256, R=8, P=3 probes, 10 steps
    C, d, _ = cases.lowrank_diag(511, 2, 256, 8, 1)
    V = cases.randn(512, 2, 256, 3, dtype=np.float32)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    q3, t3 = lanczos_tridiag(A._matmul, max_iter=10, dtype=torch.float32, device=torch.device("cpu"),
                             matrix_shape=A.matrix_shape, batch_shape=A.batch_shape, init_vecs=T(V))
    out.update(q_batch=q3, t_batch=t3)
    save("g5_lanczos", checksum=cases.checksum(M, v0, v2, C, d, V), **out)


# ------------------------------------------------------------------------------------------------
def g6_matmuls():
    print("G6 _matmul of the hot-path operators")
    out = {}
    C, d, v = cases.lowrank_diag(601, 3, 256, 8, 5)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    out["y_lowrank_diag"] = A._matmul(T(v))
    out["y_lowrank"] = LowRankRootLinearOperator(T(C))._matmul(T(v))
    out["y_diag"] = DiagLinearOperator(T(d))._matmul(T(v))
    sig = np.array([[0.25], [0.5], [2.0]], dtype=np.float32)
    A2 = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), ConstantDiagLinearOperator(T(sig), 256))
    out["y_lowrank_constdiag"] = A2._matmul(T(v))
    # broadcast batch: unbatched operator times batched rhs
    out["y_lowrank_diag_bcast"] = AddedDiagLinearOperator(
        LowRankRootLinearOperator(T(C[0])), DiagLinearOperator(T(d[0])))._matmul(T(v))
    K, dd, vv = cases.dense_diag(611, 2, 96, 3)
    out["y_dense_diag"] = AddedDiagLinearOperator(DenseLinearOperator(T(K)), DiagLinearOperator(T(dd)))._matmul(T(vv))
    out["y_dense"] = DenseLinearOperator(T(K))._matmul(T(vv))
    K1, K2, s, vk = cases.kron_factors(621, 2, 12, 20, 3)
    kp = KroneckerProductLinearOperator(T(K1), T(K2))
    out["y_kron"] = kp._matmul(T(vk))
    out["y_kron_diag"] = AddedDiagLinearOperator(kp, ConstantDiagLinearOperator(T(s), 240))._matmul(T(vk))
    out["diag_kron"] = kp._diagonal()
    save("g6_matmul", checksum=cases.checksum(C, d, v, K, dd, vv, K1, K2, vk), **out)


def g7_low_rank_root_added_diag():
    """`LowRankRoot + Diag` routes to LowRankRootAddedDiagLinearOperator, whose solve / logdet are
    the Woodbury closed forms (low_rank_root_added_diag_linear_operator.py:36-101), no CG."""
    from linear_operator.operators import LowRankRootAddedDiagLinearOperator

    C, d, rhs = cases.lowrank_diag(701, 3, 1024, 16, 3)
    A = LowRankRootLinearOperator(T(C)) + DiagLinearOperator(T(d))
    assert isinstance(A, LowRankRootAddedDiagLinearOperator)
    out = {"x": A.solve(T(rhs)), "logdet": A.logdet(), "chol_cap_mat": A.chol_cap_mat}
    iq, ld = A.inv_quad_logdet(T(rhs), logdet=True)
    out["inv_quad"], out["iq_logdet"] = iq, ld
    iqn, _ = A.inv_quad_logdet(T(rhs), logdet=False, reduce_inv_quad=False)
    out["inv_quad_noreduce"] = iqn
    sig = np.array([[0.3], [0.7], [1.1]], dtype=np.float32)
    Ac = LowRankRootLinearOperator(T(C)).add_diagonal(T(sig))
    assert isinstance(Ac, LowRankRootAddedDiagLinearOperator)
    out["x_const"], out["logdet_const"] = Ac.solve(T(rhs)), Ac.logdet()
    # fp64 reference values of the same closed form
    C64, d64, r64 = C.astype(np.float64), d.astype(np.float64), rhs.astype(np.float64)
    dense = C64 @ np.swapaxes(C64, -1, -2) + np.stack([np.diag(x) for x in d64])
    out["x_exact"] = np.linalg.solve(dense, r64)
    out["logdet_exact"] = np.linalg.slogdet(dense)[1]
    save("g7_lowrank_added_diag", checksum=cases.checksum(C, d, rhs, sig), **out)


def g8_root_decomposition():
    """RootDecomposition.forward (Lanczos consumers) with SUPPLIED initial vectors (the default
    randn start is not reproducible), root_inv_decomposition's choice among several initial vectors, and the dense
    Cholesky-method root for a small operator."""
    from linear_operator.functions._root_decomposition import RootDecomposition

    C, d, _ = cases.lowrank_diag(801, 2, 512, 8, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    v1 = cases.randn(802, 2, 512, 1, dtype=np.float32)
    v3 = cases.randn(803, 2, 512, 3, dtype=np.float32)
    tv = cases.randn(804, 2, 512, 2, dtype=np.float32)
    out = {}
    for name, iv in (("p1", v1), ("p3", v3)):
        root, inv = RootDecomposition.apply(A.representation_tree(), 12, A.dtype, A.device, A.batch_shape,
                                            A.matrix_shape, True, True, T(iv), *A.representation())
        out[f"root_{name}"], out[f"inv_{name}"] = root, inv
        out[f"rrt_tv_{name}"] = root @ (root.mT @ T(tv))      # R R^T t   (sign-invariant)
        out[f"iit_tv_{name}"] = inv @ (inv.mT @ T(tv))
    with settings.max_cholesky_size(0):
        Rinv = A.root_inv_decomposition(initial_vectors=T(v3), test_vectors=T(tv))
    out["best_iit_tv"] = Rinv.root.to_dense() @ (Rinv.root.to_dense().mT @ T(tv))
    dense = (T(C) @ T(C).mT + torch.diag_embed(T(d)))
    out["A_tv"] = dense @ T(tv)
    out["Ainv_tv"] = np.linalg.solve(dense.double().numpy(), tv.astype(np.float64))  # (torch fp64 solve hangs here)
    save("g8_root_decomposition", checksum=cases.checksum(C, d, v1, v3, tv), **out)


def g9_backward():
    """gradients of Matmul / Solve / InvQuad / InvQuadLogdet through the reference's autograd
    Functions (`_bilinear_derivative` of Dense, Diag, ConstantDiag, Root, Sum).  N = 1024 < min_preconditioning_size, so
    CG runs unpreconditioned and the stochastic logdet gradient has no preconditioner terms; probes are injected."""
    out = {}
    C, d, rhs = cases.lowrank_diag(901, 2, 1024, 8, 3)
    W = cases.randn(902, 2, 1024, 3, dtype=np.float32)
    Z = cases.randn(903, 2, 1024, 6, dtype=np.float32)

    class Probed(AddedDiagLinearOperator):
        def _probe_vectors_and_norms(self):
            z = T(Z)
            n = z.norm(dim=-2, keepdim=True)
            return z / n, n

    def leaves():
        return [T(x).clone().requires_grad_(True) for x in (C, d, rhs)]

    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-5), settings.max_cg_iterations(200):
        Ct, dt, rt = leaves()
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        ((A @ rt) * T(W)).sum().backward()
        out["mm_dC"], out["mm_dd"], out["mm_drhs"] = Ct.grad, dt.grad, rt.grad
        Ct, dt, rt = leaves()
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        x = A.solve(rt)
        (x * T(W)).sum().backward()
        out["solve_x"], out["solve_dC"], out["solve_dd"], out["solve_drhs"] = x, Ct.grad, dt.grad, rt.grad
        Ct, dt, rt = leaves()
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        iq = A.inv_quad(rt)
        iq.sum().backward()
        out["iq"], out["iq_dC"], out["iq_dd"], out["iq_drhs"] = iq, Ct.grad, dt.grad, rt.grad
        Ct, dt, rt = leaves()
        A = Probed(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        with settings.num_trace_samples(6):
            iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        out["iql_iq"], out["iql_ld"] = iq, ld
        out["iql_dC"], out["iql_dd"], out["iql_drhs"] = Ct.grad, dt.grad, rt.grad
        # dense + constant diagonal
        Kd, _, rd = cases.dense_diag(904, 2, 300, 2)
        sig = np.array([[0.4], [0.9]], dtype=np.float32)
        Wd = cases.randn(905, 2, 300, 2, dtype=np.float32)
        Kt, st, rdt = [T(x).clone().requires_grad_(True) for x in (Kd, sig, rd)]
        Ad = AddedDiagLinearOperator(DenseLinearOperator(Kt), ConstantDiagLinearOperator(st, 300))
        xd = Ad.solve(rdt)
        (xd * T(Wd)).sum().backward()
        out["dense_x"], out["dense_dK"], out["dense_dsig"], out["dense_drhs"] = xd, Kt.grad, st.grad, rdt.grad
    # Kronecker product + constant diagonal (explicit AddedDiagLinearOperator = the CG path), 12 (x) 20
    K1, K2, sk, rk = cases.kron_factors(907, 2, 12, 20, 2)
    Wk = cases.randn(908, 2, 240, 2, dtype=np.float32)
    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-5), settings.max_cg_iterations(400):
        k1t, k2t, skt, rkt = [T(x).clone().requires_grad_(True) for x in (K1, K2, sk, rk)]
        Ak = AddedDiagLinearOperator(KroneckerProductLinearOperator(DenseLinearOperator(k1t), DenseLinearOperator(k2t)),
                                     ConstantDiagLinearOperator(skt, 240))
        ((Ak @ rkt) * T(Wk)).sum().backward()
        out["kron_mm_dK1"], out["kron_mm_dK2"], out["kron_mm_dsig"] = k1t.grad, k2t.grad, skt.grad
        k1t, k2t, skt, rkt = [T(x).clone().requires_grad_(True) for x in (K1, K2, sk, rk)]
        Ak = AddedDiagLinearOperator(KroneckerProductLinearOperator(DenseLinearOperator(k1t), DenseLinearOperator(k2t)),
                                     ConstantDiagLinearOperator(skt, 240))
        xk = Ak.solve(rkt)
        (xk * T(Wk)).sum().backward()
        out["kron_x"], out["kron_dK1"], out["kron_dK2"] = xk, k1t.grad, k2t.grad
        out["kron_dsig"], out["kron_drhs"] = skt.grad, rkt.grad
    save("g9_backward", checksum=cases.checksum(C, d, rhs, W, Z, Kd, sig, rd, Wd, K1, K2, sk, rk, Wk), **out)


def g10_backward_preconditioned():
    """InvQuadLogdet gradients WITH the pivoted-Cholesky preconditioner (N = 2048 >= min_preconditioning_size): the
    reference back-propagates through the preconditioner (PivotedCholesky.backward, the QR of _init_cache) and through
    logdet P.  R = 8 < 15: the factorisation stops at m = 8 pivots.  Probes injected."""
    C, d, rhs = cases.lowrank_diag(1001, 2, 2048, 8, 1)
    Z = cases.randn(1002, 2, 2048, 6, dtype=np.float32)

    class Probed(AddedDiagLinearOperator):
        def _probe_vectors_and_norms(self):
            z = T(Z)
            n = z.norm(dim=-2, keepdim=True)
            return z / n, n

    out = {}
    with settings.cg_tolerance(1e-5), settings.max_cg_iterations(200), settings.num_trace_samples(6):
        Ct, dt, rt = [T(x).clone().requires_grad_(True) for x in (C, d, rhs)]
        A = Probed(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        out["iq"], out["ld"], out["dC"], out["dd"], out["drhs"] = iq, ld, Ct.grad, dt.grad, rt.grad
        Ct, dt, rt = [T(x).clone().requires_grad_(True) for x in (C, d, rhs)]
        A = Probed(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        ld = A.logdet()
        (ld * T(np.array([1.5, -0.5], dtype=np.float32))).sum().backward()
        out["ld_only"], out["ld_dC"], out["ld_dd"] = ld, Ct.grad, dt.grad
    # PivotedCholesky.backward on its own: loss = sum(L o G)
    Gl = cases.randn(1003, 2, 2048, 8, dtype=np.float32)
    Ct = T(C).clone().requires_grad_(True)
    Lpc = LowRankRootLinearOperator(Ct).pivoted_cholesky(rank=15)
    (Lpc * T(Gl)).sum().backward()
    out["pc_L"], out["pc_dC"] = Lpc, Ct.grad
    # constant diagonal: sigma [2, 1]
    sig = np.array([[0.6], [1.3]], dtype=np.float32)
    with settings.cg_tolerance(1e-5), settings.max_cg_iterations(200), settings.num_trace_samples(6):
        Ct, st, rt = [T(x).clone().requires_grad_(True) for x in (C, sig, rhs)]
        A = Probed(LowRankRootLinearOperator(Ct), ConstantDiagLinearOperator(st, 2048))
        iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        out["c_iq"], out["c_ld"], out["c_dC"], out["c_dsig"], out["c_drhs"] = iq, ld, Ct.grad, st.grad, rt.grad
    # exact gradient of logdet(C C^T + D): d/dd = diag(A^-1), d/dC = 2 A^-1 C   (fp64)
    C64, d64 = C.astype(np.float64), d.astype(np.float64)
    dense = C64 @ np.swapaxes(C64, -1, -2) + np.stack([np.diag(x) for x in d64])
    Ainv = np.linalg.inv(dense)
    out["exact_dlogdet_dd"] = np.diagonal(Ainv, axis1=-2, axis2=-1)
    out["exact_dlogdet_dC"] = 2.0 * (Ainv @ C64)
    save("g10_backward_precond", checksum=cases.checksum(C, d, rhs, Z, sig, Gl), **out)


def g11_diagonalization():
    """SURVEY 8(f) rank 2, second half: Diagonalization.forward through the public `diagonalization(method="lanczos")`.
    The Function draws its Lanczos start vector with torch.randn (utils/lanczos.py:31-33) and offers no way to pass
    one, so the draw is replaced by a seeded vector for the duration of the call (stored with the outputs)."""
    from unittest import mock

    out = {}
    C, d, _ = cases.lowrank_diag(1201, 2, 384, 8, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    v0 = cases.randn(1202, 384, 1, dtype=np.float32)
    tv = cases.randn(1203, 2, 384, 3, dtype=np.float32)

    def fake_randn(*size, dtype=None, device=None, **kw):
        assert tuple(size) == (384, 1), size
        return T(v0).to(dtype=dtype, device=device)

    with mock.patch("linear_operator.utils.lanczos.torch.randn", side_effect=fake_randn), \
            settings.max_root_decomposition_size(20):
        evals, evecs = A.diagonalization(method="lanczos")
    Q = evecs.to_dense()
    out["evals"], out["evecs"] = evals, Q
    out["recon_tv"] = Q @ (evals.unsqueeze(-1) * (Q.mT @ T(tv)))  # Q diag(lambda) Q^T t  (sign-invariant)
    # complete diagonalization of a small non-batch dense matrix (any start vector gives the same spectrum)
    Kd, dd, _ = cases.dense_diag(1204, 1, 40, 1)
    M = T(Kd[0]) + torch.diag_embed(T(dd[0]))
    v1 = cases.randn(1205, 40, 1, dtype=np.float32)

    def fake_randn1(*size, dtype=None, device=None, **kw):
        return T(v1).to(dtype=dtype, device=device)

    with mock.patch("linear_operator.utils.lanczos.torch.randn", side_effect=fake_randn1):
        e2, q2 = DenseLinearOperator(M).diagonalization(method="lanczos")
    out["dense_evals"], out["dense_evecs"] = e2, q2.to_dense()
    out["dense_M"] = M
    e3, q3 = DenseLinearOperator(M).diagonalization(method="symeig")
    out["symeig_evals"] = e3
    save("g11_diagonalization", checksum=cases.checksum(C, d, v0, tv, Kd, dd, v1), v0=v0, v1=v1, **out)


def g12_kronecker_added_diag():
    """SURVEY 8(f) rank 3, second half: the operator the reference's default routing builds for
    `KroneckerProduct + ConstantDiag` / `.add_diagonal(sigma2)`: eigendecomposition closed forms for solve, logdet and
    inv_quad_logdet, with the gradients the reference's autograd yields (N = 24 * 36 = 864 > max_cholesky_size)."""
    from linear_operator.operators import KroneckerProductAddedDiagLinearOperator

    K1, K2, _, rhs = cases.kron_factors(1301, 2, 24, 36, 3)
    sig = np.array([[0.3], [0.05]], dtype=np.float32)
    W = cases.randn(1302, 2, 864, 3, dtype=np.float32)
    out = {}

    def leaves():
        return [T(x).clone().requires_grad_(True) for x in (K1, K2, sig, rhs)]

    k1, k2, st, rt = leaves()
    A = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(k2)) + ConstantDiagLinearOperator(st, 864)
    assert isinstance(A, KroneckerProductAddedDiagLinearOperator)
    x = A.solve(rt)
    (x * T(W)).sum().backward()
    out["x"], out["x_dK1"], out["x_dK2"], out["x_dsig"], out["x_drhs"] = x, k1.grad, k2.grad, st.grad, rt.grad
    k1, k2, st, rt = leaves()
    A = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(k2)).add_diagonal(st)
    assert isinstance(A, KroneckerProductAddedDiagLinearOperator)
    iq, ld = A.inv_quad_logdet(rt, logdet=True)
    (iq.sum() + (ld * T(np.array([1.5, -0.5], dtype=np.float32))).sum()).backward()
    out["iq"], out["ld"] = iq, ld
    out["iql_dK1"], out["iql_dK2"], out["iql_dsig"], out["iql_drhs"] = k1.grad, k2.grad, st.grad, rt.grad
    dense = torch.stack([torch.kron(T(K1)[i].double(), T(K2)[i].double()) for i in range(2)])
    dense = dense + torch.diag_embed(T(sig).double().expand(2, 864))
    out["x_exact"] = np.linalg.solve(dense.numpy(), rhs.astype(np.float64))
    out["ld_exact"] = np.linalg.slogdet(dense.numpy())[1]
    save("g12_kron_added_diag", checksum=cases.checksum(K1, K2, sig, rhs, W), **out)


def g13_minres():
    """linear_operator.utils.minres.minres -- several shifts at once, the (value = -1, per-member
    shifts) form contour_integral_quad calls it with, a vector right-hand side without shifts, the Woodbury
    preconditioner closure of AddedDiagLinearOperator, and an all-zero column."""
    from linear_operator.utils.minres import minres

    C, d, rhs = cases.lowrank_diag(1401, 2, 300, 8, 3)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    mm = lambda v: A._matmul(v)  # noqa: E731
    sh = np.array([0.0, 0.5, 3.0], dtype=np.float32)
    sh2 = -np.array([[0.0, 0.0], [0.3, 0.2], [2.0, 5.0]], dtype=np.float32)
    out = {"sh": sh, "sh2": sh2}
    out["x_shifts"] = minres(mm, T(rhs), shifts=T(sh), max_iter=200)
    out["x_ciq"] = minres(mm, T(rhs), shifts=T(sh2), value=-1, max_iter=200)
    A0 = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C[0])), DiagLinearOperator(T(d[0])))
    out["x_vec"] = minres(lambda v: A0._matmul(v), T(rhs[0, :, 0]), max_iter=200)  # 1-D rhs, non-batch operator
    rz = rhs.copy()
    rz[1, :, 2] = 0.0
    out["x_zero_col"] = minres(mm, T(rz), shifts=T(sh[:2]), max_iter=30)
    with settings.min_preconditioning_size(0), settings.max_preconditioner_size(4):
        pre, _, _ = A._preconditioner()
        out["x_precond"] = minres(mm, T(rhs), shifts=T(sh), max_iter=200, preconditioner=pre)
    A64 = (T(C) @ T(C).mT + torch.diag_embed(T(d))).double().numpy()
    out["x_exact"] = np.stack([np.linalg.solve(A64 + s * np.eye(300), rhs.astype(np.float64)) for s in sh])
    save("g13_minres", checksum=cases.checksum(C, d, rhs), **out)


def g14_sqrt_inv_matmul():
    """contour integral quadrature -- LinearOperator.sqrt_inv_matmul (A^{-1/2} R and
    L A^{-1/2} R with the inverse quadratic form of L) forward and backward, N = 300 (no preconditioner below
    settings.min_preconditioning_size), 15 quadrature points."""
    C, d, rhs = cases.lowrank_diag(1501, 2, 300, 8, 3)
    lhs = cases.randn(1502, 2, 4, 300, dtype=np.float32)
    W = cases.randn(1503, 2, 300, 3, dtype=np.float32)
    W2 = cases.randn(1504, 2, 4, 3, dtype=np.float32)
    out = {}
    Ct, dt, rt = [T(x).clone().requires_grad_(True) for x in (C, d, rhs)]
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
    res = A.sqrt_inv_matmul(rt)
    (res * T(W)).sum().backward()
    out["res"], out["dC"], out["dd"], out["drhs"] = res, Ct.grad, dt.grad, rt.grad
    Ct, dt, rt, lt = [T(x).clone().requires_grad_(True) for x in (C, d, rhs, lhs)]
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
    res2, iq = A.sqrt_inv_matmul(rt, lt)
    ((res2 * T(W2)).sum() + (iq * T(np.array([[1.0, -0.5, 2.0, 0.3]], dtype=np.float32))).sum()).backward()
    out["l_res"], out["l_iq"] = res2, iq
    out["l_dC"], out["l_dd"], out["l_drhs"], out["l_dlhs"] = Ct.grad, dt.grad, rt.grad, lt.grad
    solves, weights, no_shift, shifts = linear_operator.utils.contour_integral_quad(
        AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d))), T(rhs), inverse=False)
    out["sqrt_res"] = (solves * weights).sum(0)  # A^{1/2} rhs
    out["shifts"], out["weights"] = shifts, weights
    A64 = (T(C) @ T(C).mT + torch.diag_embed(T(d))).double().numpy()
    ev, Q = np.linalg.eigh(A64)
    out["exact_inv_sqrt"] = (Q / np.sqrt(ev)[..., None, :]) @ np.swapaxes(Q, -1, -2) @ rhs.astype(np.float64)
    out["exact_sqrt"] = (Q * np.sqrt(ev)[..., None, :]) @ np.swapaxes(Q, -1, -2) @ rhs.astype(np.float64)
    save("g14_sqrt_inv_matmul", checksum=cases.checksum(C, d, rhs, lhs, W, W2), **out)


def g15_lanczos_consumers_backward():
    """Backward of RootDecomposition and Diagonalization (functions/_root_decomposition.py:104-171,
    functions/_diagonalization.py:62-88) with sign-invariant losses (R R^T t, R_inv R_inv^T t, Q diag(s) Q^T)."""
    from unittest import mock

    from linear_operator.functions._root_decomposition import RootDecomposition

    C, d, _ = cases.lowrank_diag(1601, 2, 256, 8, 1)
    v1 = cases.randn(1602, 2, 256, 1, dtype=np.float32)
    tv = cases.randn(1603, 2, 256, 2, dtype=np.float32)
    W1 = cases.randn(1604, 2, 256, 2, dtype=np.float32)
    W2 = cases.randn(1605, 2, 256, 2, dtype=np.float32)
    out = {}
    for name, want_inv in (("both", True), ("root", False)):
        Ct, dt = T(C).clone().requires_grad_(True), T(d).clone().requires_grad_(True)
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        root, inv = RootDecomposition.apply(A.representation_tree(), 12, A.dtype, A.device, A.batch_shape,
                                            A.matrix_shape, True, want_inv, T(v1), *A.representation())
        loss = ((root @ (root.mT @ T(tv))) * T(W1)).sum()
        if want_inv:
            loss = loss + ((inv @ (inv.mT @ T(tv))) * T(W2)).sum()
        loss.backward()
        out[f"{name}_loss"], out[f"{name}_dC"], out[f"{name}_dd"] = loss.detach(), Ct.grad, dt.grad
    Kd, dd, _ = cases.dense_diag(1606, 1, 40, 1)
    M0 = (T(Kd[0]) + torch.diag_embed(T(dd[0])))
    v2 = cases.randn(1607, 40, 1, dtype=np.float32)
    w = cases.randn(1608, 40, dtype=np.float32)
    sdiag = cases.randn(1609, 40, dtype=np.float32)
    Ws = cases.randn(1610, 40, 40, dtype=np.float32)
    M = M0.clone().requires_grad_(True)
    with mock.patch("linear_operator.utils.lanczos.torch.randn", side_effect=lambda *a, dtype=None, device=None, **k: T(v2)):
        evals, evecs = DenseLinearOperator(M).diagonalization(method="lanczos")
    q = evecs.to_dense()
    order = torch.argsort(evals)  # (fix the eigenvalue order: the weights must meet the same eigenpairs everywhere)
    evs, qs = evals[order], q[:, order]
    loss = (evs * T(w)).sum() + (((qs * T(sdiag)) @ qs.mT) * T(Ws)).sum()
    loss.backward()
    out["diag_loss"], out["diag_dM"], out["diag_M"] = loss.detach(), M.grad, M0
    save("g15_lanczos_consumers_backward", checksum=cases.checksum(C, d, v1, tv, W1, W2, Kd, dd, v2, w, sdiag, Ws),
         v2=v2, **out)


def g16_sum_operators():
    """Real sums (north_star: "their Sum/Added/Kronecker compositions"): Sum(LowRankRoot, Dense) + Diag and
    PsdSum(LowRankRoot, LowRankRoot) + Diag -- matmul (sum_linear_operator.py:47-51), pivoted Cholesky of the sum (rows
    through SumLinearOperator._get_indices :39-41), solve, inv_quad_logdet with injected probes, and the gradients."""
    print("G16 multi-term sums")
    C, d, rhs = cases.lowrank_diag(1601, 3, 2048, 16, 1)
    Kd, _, V = cases.dense_diag(1602, 3, 2048, 3)
    Kd = (Kd * np.float32(0.25)).astype(np.float32)
    Z, Zn = cases.probes(1603, 3, 2048, 6)
    S = SumLinearOperator(LowRankRootLinearOperator(T(C)), DenseLinearOperator(T(Kd)))
    A = _ProbedAddedDiag(S, DiagLinearOperator(T(d)))
    A._probes = (T(Z), T(Zn))
    out = {"mv": A._matmul(T(V)), "mv_sum_only": S._matmul(T(V))}
    L, piv = S.pivoted_cholesky(rank=15, return_pivots=True)
    out["pc_L"], out["pc_piv"] = L, piv
    with settings.cg_tolerance(1e-4):
        x, spy, w = _with_spy(lambda: A.solve(T(rhs)))
    out["x"], out["x_matvecs"], out["x_warned"] = x, spy.records[0]["matvecs"], w
    with settings.cg_tolerance(1e-4):
        (iq, ld), spy, w = _with_spy(lambda: A.inv_quad_logdet(T(rhs), logdet=True))
    solves, t_mat = spy.records[0]["out"]
    _, _, logdet_p = A._preconditioner()
    out.update(iq=iq, ld=ld, solves=solves, t_mat=t_mat, logdet_p=logdet_p, iql_matvecs=spy.records[0]["matvecs"])
    dense = T(C).double() @ T(C).double().mT + T(Kd).double() + torch.diag_embed(T(d).double())
    out["x_exact"] = np.linalg.solve(dense.numpy(), rhs.astype(np.float64)).astype(np.float32)
    out["logdet_exact"] = np.linalg.slogdet(dense.numpy())[1].astype(np.float32)
    # gradients of iq.sum() + ld.sum() through the preconditioned path (pivoted Cholesky of the SUM in the graph)
    wproj = cases.randn(1604, 2048, 2, dtype=np.float32)
    with settings.cg_tolerance(1e-5), settings.max_cg_iterations(200):
        Ct, Kt, dt, rt = [T(a).clone().requires_grad_(True) for a in (C, Kd, d, rhs)]
        Ag = _ProbedAddedDiag(SumLinearOperator(LowRankRootLinearOperator(Ct), DenseLinearOperator(Kt)),
                              DiagLinearOperator(dt))
        Ag._probes = (T(Z), T(Zn))
        iq, ld = Ag.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        out.update(g_iq=iq, g_ld=ld, g_dC=Ct.grad, g_dd=dt.grad, g_drhs=rt.grad, g_dK_proj=Kt.grad @ T(wproj),
                   g_dK_diag=torch.diagonal(Kt.grad, dim1=-2, dim2=-1))
    # PsdSum of two low-rank roots + diagonal: solve
    C2, _, _ = cases.lowrank_diag(1605, 3, 2048, 8, 1)
    P = PsdSumLinearOperator(LowRankRootLinearOperator(T(C)), LowRankRootLinearOperator(T(C2)))
    A2 = AddedDiagLinearOperator(P, DiagLinearOperator(T(d)))
    with settings.cg_tolerance(1e-4):
        x2, spy, w = _with_spy(lambda: A2.solve(T(rhs)))
    L2, piv2 = P.pivoted_cholesky(rank=15, return_pivots=True)
    out.update(psd_x=x2, psd_matvecs=spy.records[0]["matvecs"], psd_pc_L=L2, psd_pc_piv=piv2,
               psd_mv=A2._matmul(T(V)))
    save("g16_sum_operators", checksum=cases.checksum(C, d, rhs, Kd, V, Z, wproj, C2), **out)



def g17_low_rank_root_added_diag_backward():
    """Gradients of LowRankRootAddedDiagLinearOperator (the reference differentiates the torch expressions of the
    Woodbury closed forms, low_rank_root_added_diag_linear_operator.py:62-103,117-170): inv_quad + logdet, solve, and
    the constant-diagonal variant."""
    from linear_operator.operators import LowRankRootAddedDiagLinearOperator

    print("G17 LowRankRootAddedDiag gradients")
    C, d, rhs = cases.lowrank_diag(1701, 3, 1024, 16, 3)
    W = cases.randn(1702, 3, 1024, 3, dtype=np.float32)
    out = {}
    Ct, dt, rt = [T(a).clone().requires_grad_(True) for a in (C, d, rhs)]
    A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
    assert isinstance(A, LowRankRootAddedDiagLinearOperator)
    iq, ld = A.inv_quad_logdet(rt, logdet=True)
    (iq.sum() + ld.sum()).backward()
    out.update(iq=iq, ld=ld, dC=Ct.grad, dd=dt.grad, drhs=rt.grad)
    Ct, dt, rt = [T(a).clone().requires_grad_(True) for a in (C, d, rhs)]
    A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
    x = A.solve(rt)
    (x * T(W)).sum().backward()
    out.update(s_x=x, s_dC=Ct.grad, s_dd=dt.grad, s_drhs=rt.grad)
    sig = np.array([[0.3], [0.7], [1.1]], dtype=np.float32)
    Ct, st = T(C).clone().requires_grad_(True), T(sig).clone().requires_grad_(True)
    Ac = LowRankRootLinearOperator(Ct).add_diagonal(st)
    assert isinstance(Ac, LowRankRootAddedDiagLinearOperator)
    ldc = Ac.logdet()
    (ldc * T(np.array([1.0, -2.0, 0.5], dtype=np.float32))).sum().backward()
    out.update(c_ld=ldc, c_dC=Ct.grad, c_dsig=st.grad)
    save("g17_lowrank_added_diag_backward", checksum=cases.checksum(C, d, rhs, W, sig), **out)


def g18_low_rank_root_added_diag_wide_root():
    """LowRankRootAddedDiagLinearOperator with a root of rank 48 (> 32): solve, inv_quad_logdet and their gradients
    (low_rank_root_added_diag_linear_operator.py:62-103,117-170)."""
    from linear_operator.operators import LowRankRootAddedDiagLinearOperator

    print("G18 LowRankRootAddedDiag, rank 48")
    C, d, rhs = cases.lowrank_diag(1801, 2, 768, 48, 2)
    W = cases.randn(1802, 2, 768, 2, dtype=np.float32)
    out = {}
    Ct, dt, rt = [T(a).clone().requires_grad_(True) for a in (C, d, rhs)]
    A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
    assert isinstance(A, LowRankRootAddedDiagLinearOperator)
    iq, ld = A.inv_quad_logdet(rt, logdet=True)
    (iq.sum() + ld.sum()).backward()
    out.update(iq=iq, ld=ld, dC=Ct.grad, dd=dt.grad, drhs=rt.grad)
    Ct, dt, rt = [T(a).clone().requires_grad_(True) for a in (C, d, rhs)]
    A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
    x = A.solve(rt)
    (x * T(W)).sum().backward()
    out.update(s_x=x, s_dC=Ct.grad, s_dd=dt.grad, s_drhs=rt.grad)
    save("g18_lowrank_added_diag_rank48", checksum=cases.checksum(C, d, rhs, W), **out)

def g19_kronecker_three_factors():
    """KroneckerProductLinearOperator of THREE dense factors (kronecker_product_linear_operator.py:34-45, 272-284):
    matmul, and `AddedDiag(kron, Diag)` solve / inv_quad_logdet with the reference's autograd gradients for the three
    factors (N = 6 * 8 * 10 = 480 <= max_cholesky_size: the reference's exact Cholesky route)."""
    print("G19 Kronecker, three factors")
    K1, K2, _, rhs = cases.kron_factors(1901, 2, 6, 8, 3)
    K3, _, _, _ = cases.kron_factors(1902, 2, 10, 2, 1)
    rhs = cases.randn(1903, 2, 480, 3, dtype=np.float32)
    d = (np.abs(cases.randn(1904, 2, 480, dtype=np.float32)) * 0.2 + 0.3).astype(np.float32)
    W = cases.randn(1905, 2, 480, 3, dtype=np.float32)
    out = {}

    def leaves():
        return [T(x).clone().requires_grad_(True) for x in (K1, K2, K3, d, rhs)]

    k1, k2, k3, dt, rt = leaves()
    Kp = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(k2), DenseLinearOperator(k3))
    out["mm"] = Kp.matmul(rt)
    A = AddedDiagLinearOperator(Kp, DiagLinearOperator(dt))
    x = A.solve(rt)
    (x * T(W)).sum().backward()
    out.update(x=x, x_dK1=k1.grad, x_dK2=k2.grad, x_dK3=k3.grad, x_dd=dt.grad, x_drhs=rt.grad)
    k1, k2, k3, dt, rt = leaves()
    Kp = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(k2), DenseLinearOperator(k3))
    A = AddedDiagLinearOperator(Kp, DiagLinearOperator(dt))
    iq, ld = A.inv_quad_logdet(rt, logdet=True)
    (iq.sum() + (ld * T(np.array([1.5, -0.5], dtype=np.float32))).sum()).backward()
    out.update(iq=iq, ld=ld, iql_dK1=k1.grad, iql_dK2=k2.grad, iql_dK3=k3.grad, iql_dd=dt.grad, iql_drhs=rt.grad)
    dense = torch.stack([torch.kron(torch.kron(T(K1)[i].double(), T(K2)[i].double()), T(K3)[i].double()) for i in range(2)])
    out["mm_exact"] = (dense @ T(rhs).double()).numpy()
    dense = dense + torch.diag_embed(T(d).double())
    out["x_exact"] = np.linalg.solve(dense.numpy(), rhs.astype(np.float64))
    out["ld_exact"] = np.linalg.slogdet(dense.numpy())[1]
    save("g19_kron_three_factors", checksum=cases.checksum(K1, K2, K3, d, rhs, W), **out)

def g20_kronecker_structured_diag():
    """KroneckerProduct + KroneckerProductDiag (kronecker_product_added_diag_linear_operator.py:94-128, 166-219): the
    Woodbury / symmetrisation solve and the structured logdet, for full per-factor diagonals and for constant factors,
    with the reference's autograd gradients (max_cholesky_size(0): the structured branches, not the dense Cholesky)."""
    import linear_operator
    from linear_operator.operators import KroneckerProductAddedDiagLinearOperator, KroneckerProductDiagLinearOperator

    print("G20 Kronecker + Kronecker-structured diagonal")
    K1, K2, _, _ = cases.kron_factors(2001, 2, 6, 8, 3)
    rhs = cases.randn(2002, 2, 48, 3, dtype=np.float32)
    W = cases.randn(2003, 2, 48, 3, dtype=np.float32)
    d1 = (np.abs(cases.randn(2004, 2, 6, dtype=np.float32)) * 0.3 + 0.4).astype(np.float32)
    d2 = (np.abs(cases.randn(2005, 2, 8, dtype=np.float32)) * 0.3 + 0.4).astype(np.float32)
    c1 = np.array([[0.6], [0.9]], dtype=np.float32)
    c2 = np.array([[0.5], [0.3]], dtype=np.float32)
    wld = T(np.array([1.5, -0.5], dtype=np.float32))
    out = {}
    with linear_operator.settings.max_cholesky_size(0):
        for tag, (a, b) in (("full", (d1, d2)), ("const", (c1, c2))):
            def leaves():
                return [T(x).clone().requires_grad_(True) for x in (K1, K2, a, b, rhs)]

            def build(k1, k2, ta, tb):
                if tag == "full":
                    D = KroneckerProductDiagLinearOperator(DiagLinearOperator(ta), DiagLinearOperator(tb))
                else:
                    D = KroneckerProductDiagLinearOperator(ConstantDiagLinearOperator(ta, 6), ConstantDiagLinearOperator(tb, 8))
                A = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(k2)) + D
                assert isinstance(A, KroneckerProductAddedDiagLinearOperator)
                return A

            k1, k2, ta, tb, rt = leaves()
            x = build(k1, k2, ta, tb).solve(rt)
            (x * T(W)).sum().backward()
            out.update({f"{tag}_x": x, f"{tag}_x_dK1": k1.grad, f"{tag}_x_dK2": k2.grad, f"{tag}_x_da": ta.grad,
                        f"{tag}_x_db": tb.grad, f"{tag}_x_drhs": rt.grad})
            k1, k2, ta, tb, rt = leaves()
            if tag == "full":
                iq, ld = build(k1, k2, ta, tb).inv_quad_logdet(rt, logdet=True)
                (iq.sum() + (ld * wld).sum()).backward()
                out[f"{tag}_ld"] = ld
            else:  # the reference's constant-factor logdet branch (:100-110) raises AttributeError (evals is a Tensor
                # there); only the inverse quadratic form is recorded, the logdet is checked against the dense value
                iq, _ = build(k1, k2, ta, tb).inv_quad_logdet(rt, logdet=False)
                iq.sum().backward()
            out.update({f"{tag}_iq": iq, f"{tag}_iql_dK1": k1.grad, f"{tag}_iql_dK2": k2.grad,
                        f"{tag}_iql_da": ta.grad, f"{tag}_iql_db": tb.grad, f"{tag}_iql_drhs": rt.grad})
            da = a if tag == "full" else np.broadcast_to(a, (2, 6))
            db = b if tag == "full" else np.broadcast_to(b, (2, 8))
            dense = np.stack([np.kron(K1[i].astype(np.float64), K2[i].astype(np.float64))
                              + np.diag(np.kron(da[i].astype(np.float64), db[i].astype(np.float64))) for i in range(2)])
            out[f"{tag}_x_exact"] = np.linalg.solve(dense, rhs.astype(np.float64))
            out[f"{tag}_ld_exact"] = np.linalg.slogdet(dense)[1]
    save("g20_kron_structured_diag", checksum=cases.checksum(K1, K2, rhs, W, d1, d2, c1, c2), **out)


def g21_minres_fp64():
    """The reference's own test/utils/test_minres.py:17-80 recipes: float64 operands, value = -1, minres_tolerance
    1e-6, vector / matrix / batched right-hand sides, with and without the shifts [0, 1, 2], batched and unbatched
    matrices."""
    from linear_operator.utils.minres import minres

    sh = np.array([0.0, 1.0, 2.0])
    out = {"sh": sh}
    inputs = []
    runs = [("vec", (20,), (), None), ("vec_shifts", (5,), (), sh), ("mat", (20, 5), (), None),
            ("bmat", (3, 20, 5), (), None), ("bmat_bop", (3, 20, 5), (3,), None), ("mat_bop", (20, 5), (3,), None),
            ("mat_shifts", (20, 5), (), sh), ("bmat_bop_shifts", (3, 20, 5), (3,), sh), ("mat_bop_shifts", (20, 5), (3,), sh)]
    for i, (tag, rshape, mbatch, shifts) in enumerate(runs):
        size = rshape[-2] if len(rshape) > 1 else rshape[-1]
        M = cases.spd_test_matrix(2100 + i, size, batch=mbatch)
        b = cases.randn(2150 + i, *rshape)
        inputs += [M, b]
        with settings.minres_tolerance(1e-6):
            x = minres(T(M), rhs=T(b), value=-1, shifts=None if shifts is None else T(shifts))
        out[f"x_{tag}"] = x
    save("g21_minres_fp64", checksum=cases.checksum(*inputs), **out)


def g22_kronecker_iteration_pinned():
    """cfg4 with the ITERATION COUNT pinned: AddedDiag(Kron(K1, K2), ConstantDiag).solve first at cg_tolerance 1e-3 (the
    stop-rule run: `its` iterations), then again with cg_tolerance 0 and max_cg_iterations = its -- the very same
    iterates, no stop decision involved -- so that a kernel path can be compared column by column at 1e-4 after
    exactly the reference's iterations (linear_cg.py:302-308 never fires).  Two sizes: 48 (x) 48 (N = 2304) and
    128 (x) 128 (N = 16384: the matrix-core Kronecker GEMMs and the single-pass preconditioner apply)."""
    print("G22 Kronecker solve, iteration-pinned")
    out, inputs = {}, []
    for tag, seed, n in (("n48", 421, 48), ("n128", 2201, 128)):
        K1, K2, sig, rhs = cases.kron_factors(seed, 2, n, n, 1)
        N = n * n
        A = AddedDiagLinearOperator(KroneckerProductLinearOperator(T(K1), T(K2)), ConstantDiagLinearOperator(T(sig), N))
        with settings.cg_tolerance(1e-3):
            x_tol, spy, w = _with_spy(lambda: A.solve(T(rhs)))
        its = int(spy.records[0]["matvecs"]) - 1
        A2 = AddedDiagLinearOperator(KroneckerProductLinearOperator(T(K1), T(K2)), ConstantDiagLinearOperator(T(sig), N))
        with settings.cg_tolerance(0.0), settings.max_cg_iterations(its):
            x_pin, spy2, w2 = _with_spy(lambda: A2.solve(T(rhs)))
        assert int(spy2.records[0]["matvecs"]) - 1 == its
        # exact solution by the per-factor eigendecomposition in fp64: (K1 (x) K2 + s I)^-1 b
        xs = []
        for i in range(2):
            l1, q1 = np.linalg.eigh(K1[i].astype(np.float64))
            l2, q2 = np.linalg.eigh(K2[i].astype(np.float64))
            Bm = rhs[i, :, 0].astype(np.float64).reshape(n, n)
            Y = q1.T @ Bm @ q2
            Y = Y / (np.outer(l1, l2) + float(sig[i, 0]))
            xs.append((q1 @ Y @ q2.T).reshape(N, 1))
        out.update({f"x_tol_{tag}": x_tol, f"x_pinned_{tag}": x_pin, f"iterations_{tag}": its,
                    f"x_exact_{tag}": np.stack(xs).astype(np.float32), f"warned_pinned_{tag}": w2})
        inputs += [K1, K2, sig, rhs]
        d1 = float((x_pin - x_tol).norm() / x_tol.norm())
        print(f"  {tag}: {its} iterations, |x_pinned - x_tol| / |x_tol| = {d1:.2e}")
    save("g22_kron_iteration_pinned", checksum=cases.checksum(*inputs), **out)


def _tridiag_valid_len(t32, t64, rtol):
    """Per (column, member): the largest k such that the leading k x k blocks of the two tridiagonals agree ENTRY BY
    ENTRY, |a - b| <= rtol |b| + 1e-6 max |b| over the block (the absolute part only forgives off-diagonals that
    have decayed to rounding level); the block also ends where the fp64 off-diagonal has decayed below 1e-5 of the
    largest entry: the column has converged, what follows is 0 / 0 in every fp32 run."""
    nt, B = t64.shape[:2]
    K = min(t32.shape[-1], t64.shape[-1])
    out = np.zeros((nt, B), dtype=np.int64)
    for i in range(nt):
        for b in range(B):
            for m in range(1, K + 1):
                a, c = t32[i, b, :m, :m].astype(np.float64), t64[i, b, :m, :m].astype(np.float64)
                if m > 1 and abs(c[m - 1, m - 2]) <= 1e-5 * np.abs(c).max():
                    break  # the fp64 recurrence has decoupled here (column converged): later rows are not defined
                if np.any(np.abs(a - c) > rtol * np.abs(c) + 1e-6 * np.abs(c).max()):
                    break
                out[i, b] = m
    return out


def g23_tridiag_divergence_and_tight_logdet():
    """(a) For every fp32 golden that carries CG tridiagonals, the SAME reference run in fp64 on the same (fp32-valued)
    inputs, and the index up to which the reference's own fp32 run follows it (leading block, every entry to 1e-4
    relative): beyond that index the fp32 recurrence (linear_cg.py:311-332) is rounding noise in the reference
    itself, up to it a kernel path has to agree entry by entry.
    (b) Two WELL-CONDITIONED inv_quad_logdet cases with injected probes (spectrum of the preconditioned operator in
    [1, 4], |logdet| ~ 1e3): there the reference's fp32 result is within 2e-5 of its own fp64 result, so a kernel path
    can be held to rtol 1e-4 with atol 0 (functions/_inv_quad_logdet.py:112-153, test/functions/test_inv_quad_logdet.py:17-85)."""
    print("G23 tridiagonal divergence index + tight logdet")
    out, inputs = {}, []
    # ---- (a1) g1_cg_fp32_lowrank, tolerance 1, n_tridiag 4 (unpreconditioned, 20 x 20)
    C, d, rhs = cases.lowrank_diag(141, 4, 512, 8, 5)
    A32 = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    A64 = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C).double()), DiagLinearOperator(T(d).double()))
    (_, t32), _, _ = run_cg(A32._matmul, T(rhs), tolerance=1.0, n_tridiag=4)
    (x64, t64), n64, _ = run_cg(A64._matmul, T(rhs).double(), tolerance=1.0, n_tridiag=4)
    out.update(g1_t_mat_f64=t64, g1_x_f64=x64, g1_matvecs_f64=n64,
               g1_valid=_tridiag_valid_len(t32.numpy(), t64.numpy(), 1e-4))
    print("  g1 lowrank: valid rows", out["g1_valid"].min(), "..", out["g1_valid"].max(), "of", t64.shape[-1])
    inputs += [C, d, rhs]

    # ---- (a2) g4_iql_lowrank and g4_iql_dense (preconditioned, injected probes)
    def iql(make_op, rhs, Z, Zn, dt, tol=1e-4):
        A = make_op(dt)
        A._probes = (T(Z).to(dt), T(Zn).to(dt))
        with settings.cg_tolerance(tol):
            (iq, ld), spy, w = _with_spy(lambda: A.inv_quad_logdet(T(rhs).to(dt), logdet=True))
        solves, t_mat = spy.records[0]["out"]
        return iq, ld, solves, t_mat, spy.records[0]["matvecs"]

    C, d, rhs = cases.lowrank_diag(411, 3, 2048, 16, 1)
    Z, Zn = cases.probes(412, 3, 2048, 8)
    mk = lambda dt: _ProbedAddedDiag(LowRankRootLinearOperator(T(C).to(dt)), DiagLinearOperator(T(d).to(dt)))  # noqa: E731
    _, ld32, _, t32, _ = iql(mk, rhs, Z, Zn, torch.float32)
    iq64, ld64, s64, t64, n64 = iql(mk, rhs, Z, Zn, torch.float64)
    out.update(iql_lowrank_t_mat_f64=t64, iql_lowrank_logdet_f64=ld64, iql_lowrank_inv_quad_f64=iq64,
               iql_lowrank_solves_f64=s64, iql_lowrank_matvecs_f64=n64,
               iql_lowrank_valid=_tridiag_valid_len(t32.numpy(), t64.numpy(), 1e-4))
    print("  iql lowrank: valid", out["iql_lowrank_valid"].min(), "..", out["iql_lowrank_valid"].max(),
          "fp32 logdet", ld32.tolist(), "fp64", ld64.tolist())
    inputs += [C, d, rhs, Z]
    Kd, d, rhs = cases.dense_diag(431, 2, 2048, 1)
    Z, Zn = cases.probes(432, 2, 2048, 4)
    mk = lambda dt: _ProbedAddedDiag(DenseLinearOperator(T(Kd).to(dt)), DiagLinearOperator(T(d).to(dt)))  # noqa: E731
    _, ld32, _, t32, _ = iql(mk, rhs, Z, Zn, torch.float32)
    iq64, ld64, s64, t64, n64 = iql(mk, rhs, Z, Zn, torch.float64)
    out.update(iql_dense_t_mat_f64=t64, iql_dense_logdet_f64=ld64, iql_dense_inv_quad_f64=iq64,
               iql_dense_matvecs_f64=n64, iql_dense_valid=_tridiag_valid_len(t32.numpy(), t64.numpy(), 1e-4))
    print("  iql dense: valid", out["iql_dense_valid"].min(), "..", out["iql_dense_valid"].max(),
          "fp32 logdet", ld32.tolist(), "fp64", ld64.tolist())
    inputs += [Kd, d, rhs, Z]

    # ---- (b) well-conditioned logdet: d in [1.5, 2.5), C = 0.05 randn (C C^T has R eigenvalues ~ 0.0025 N)
    for tag, seed, B, N, R, P in (("wc_nopre", 2301, 3, 1024, 8, 8), ("wc_pre", 2311, 3, 2304, 32, 8)):
        g = np.random.default_rng(seed)
        C = (0.05 * g.standard_normal((B, N, R))).astype(np.float32)
        d = (g.random((B, N)) + 1.5).astype(np.float32)
        rhs = g.standard_normal((B, N, 1)).astype(np.float32)
        Z, Zn = cases.probes(seed + 1, B, N, P)
        mk = lambda dt: _ProbedAddedDiag(LowRankRootLinearOperator(T(C).to(dt)), DiagLinearOperator(T(d).to(dt)))  # noqa: E731
        iq32, ld32, s32, t32, n32 = iql(mk, rhs, Z, Zn, torch.float32)
        iq64, ld64, s64, t64, n64 = iql(mk, rhs, Z, Zn, torch.float64)
        dense = (T(C).double() @ T(C).double().mT) + torch.diag_embed(T(d).double())
        exact = np.linalg.slogdet(dense.numpy())[1]
        rel = float(((ld32.double() - ld64).abs() / ld64.abs()).max())
        print(f"  {tag}: matvecs {n32}/{n64}, t_mat {tuple(t32.shape)}, logdet fp32 {ld32.tolist()} fp64 {ld64.tolist()} "
              f"exact {exact.tolist()}; fp32 vs fp64 rel {rel:.2e}")
        assert rel < 2e-5, "not a well-conditioned case: the reference's own fp32 run is off"
        out.update({f"{tag}_logdet": ld32, f"{tag}_inv_quad": iq32, f"{tag}_solves": s32, f"{tag}_t_mat": t32,
                    f"{tag}_matvecs": n32, f"{tag}_logdet_f64": ld64, f"{tag}_inv_quad_f64": iq64,
                    f"{tag}_t_mat_f64": t64, f"{tag}_logdet_exact": exact,
                    f"{tag}_valid": _tridiag_valid_len(t32.numpy(), t64.numpy(), 1e-4)})
        inputs += [C, d, rhs, Z]
    save("g23_tridiag_divergence_tight_logdet", checksum=cases.checksum(*inputs), **out)


def g24_kronecker_256_iteration_pinned():
    """cfg4 at its real factor size, 256 (x) 256 (N = 65536), B = 2, iteration-pinned like g22: the reference at
    cg_tolerance 1e-3, then tolerance 0 with max_cg_iterations = the count it needed.  Only the pinned iterate, the
    exact solution and the iteration count are stored (fp32, 2 x 65536 each)."""
    print("G24 Kronecker 256 (x) 256, iteration-pinned")
    n, B = 256, 2
    N = n * n
    K1, K2, sig, rhs = cases.kron_factors(2401, B, n, n, 1)
    A = AddedDiagLinearOperator(KroneckerProductLinearOperator(T(K1), T(K2)), ConstantDiagLinearOperator(T(sig), N))
    with settings.cg_tolerance(1e-3):
        x_tol, spy, w = _with_spy(lambda: A.solve(T(rhs)))
    its = int(spy.records[0]["matvecs"]) - 1
    A2 = AddedDiagLinearOperator(KroneckerProductLinearOperator(T(K1), T(K2)), ConstantDiagLinearOperator(T(sig), N))
    with settings.cg_tolerance(0.0), settings.max_cg_iterations(its):
        x_pin, spy2, w2 = _with_spy(lambda: A2.solve(T(rhs)))
    assert int(spy2.records[0]["matvecs"]) - 1 == its
    xs = []
    for i in range(B):
        l1, q1 = np.linalg.eigh(K1[i].astype(np.float64))
        l2, q2 = np.linalg.eigh(K2[i].astype(np.float64))
        Y = q1.T @ rhs[i, :, 0].astype(np.float64).reshape(n, n) @ q2
        xs.append((q1 @ (Y / (np.outer(l1, l2) + float(sig[i, 0]))) @ q2.T).reshape(N, 1))
    xe = np.stack(xs)
    err_pin = float(np.abs(x_pin.numpy() - xe).max() / np.abs(xe).max())
    print(f"  {its} iterations; |x_pinned - x_tol| / |x_tol| = {float((x_pin - x_tol).norm() / x_tol.norm()):.2e}; "
          f"pinned vs exact (max-norm) {err_pin:.2e}")
    save("g24_kron256_iteration_pinned", x_pinned=x_pin, iterations=its, warned_pinned=w2,
         x_exact=xe.astype(np.float32), checksum=cases.checksum(K1, K2, sig, rhs))


def g25_fp64_preconditioned_path():
    """The preconditioned path in float64 (the reference is dtype-generic; GPyTorch users often run in double):
    pivoted Cholesky of a low-rank root and of a dense matrix, `A.solve` and `inv_quad_logdet` with injected probes of
    AddedDiag(LowRankRoot, Diag) above `min_preconditioning_size` (N = 2048)."""
    print("G25 float64: pivoted Cholesky, preconditioned solve, inv_quad_logdet")
    C, d, rhs = cases.lowrank_diag(2501, 2, 2048, 16, 2, dtype=np.float64)
    Z, Zn = cases.probes(2502, 2, 2048, 6, dtype=np.float64)
    Lr, pr = LowRankRootLinearOperator(T(C)).pivoted_cholesky(15, return_pivots=True)
    Kd, _, _ = cases.dense_diag(2503, 2, 300, 1, dtype=np.float64)
    Ld, pd_ = DenseLinearOperator(T(Kd)).pivoted_cholesky(10, return_pivots=True)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    with settings.cg_tolerance(1e-8):
        x, spy, w = _with_spy(lambda: A.solve(T(rhs)))
    Ap = _ProbedAddedDiag(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    Ap._probes = (T(Z), T(Zn))
    with settings.cg_tolerance(1e-8):
        (iq, ld), spy2, w2 = _with_spy(lambda: Ap.inv_quad_logdet(T(rhs), logdet=True))
    _, _, logdet_p = Ap._preconditioner()
    dense = (T(C) @ T(C).mT) + torch.diag_embed(T(d))
    # constant diagonal
    dc = np.full_like(d, 0.7)
    Ac = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), ConstantDiagLinearOperator(T(dc[:, :1].copy()), 2048))
    with settings.cg_tolerance(1e-8):
        xc, spyc, wc = _with_spy(lambda: Ac.solve(T(rhs)))
    save("g25_fp64_preconditioned", L_root=Lr, piv_root=pr, L_dense=Ld, piv_dense=pd_, x=x,
         matvecs=spy.records[0]["matvecs"], inv_quad=iq, logdet=ld, iql_matvecs=spy2.records[0]["matvecs"],
         logdet_p=logdet_p, logdet_exact=np.linalg.slogdet(dense.numpy())[1], x_exact=np.linalg.solve(dense.numpy(), rhs),
         x_const=xc, matvecs_const=spyc.records[0]["matvecs"], checksum=cases.checksum(C, d, rhs, Z, Kd))
    print(f"  solve matvecs {spy.records[0]['matvecs']}, iql matvecs {spy2.records[0]['matvecs']}, const {spyc.records[0]['matvecs']}; "
          f"logdet {ld.tolist()} exact {np.linalg.slogdet(dense.numpy())[1].tolist()}")


def g26_lanczos_fp64():
    """lanczos_tridiag and RootDecomposition.forward in float64: the test_lanczos.py:42-48 recipe at N = 100 run to
    completion, a batched dense operator with three start vectors stopped after 16 steps, and the root / inverse root
    of AddedDiag(LowRankRoot, Diag) from 14 steps."""
    from linear_operator.functions._root_decomposition import RootDecomposition

    print("G26 float64 lanczos_tridiag / RootDecomposition")
    dev = torch.device("cpu")
    M = cases.spd_test_matrix(2601, 100, dtype=np.float64, jitter=1e-6)
    v0 = cases.randn(2602, 100, 1, dtype=np.float64)
    q, t = lanczos_tridiag(T(M).matmul, max_iter=100, dtype=torch.float64, device=dev, matrix_shape=M.shape,
                           init_vecs=T(v0))
    Kd, _, _ = cases.dense_diag(2603, 2, 300, 1, dtype=np.float64)
    V = cases.randn(2604, 2, 300, 3, dtype=np.float64)
    qb, tb = lanczos_tridiag(T(Kd).matmul, max_iter=16, dtype=torch.float64, device=dev, matrix_shape=Kd.shape[-2:],
                             batch_shape=torch.Size([2]), init_vecs=T(V))
    C, d, _ = cases.lowrank_diag(2605, 2, 512, 8, 1, dtype=np.float64)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    v1 = cases.randn(2606, 2, 512, 1, dtype=np.float64)
    tv = cases.randn(2607, 2, 512, 2, dtype=np.float64)
    root, inv = RootDecomposition.apply(A.representation_tree(), 14, A.dtype, A.device, A.batch_shape, A.matrix_shape,
                                        True, True, T(v1), *A.representation())
    dense = T(C) @ T(C).mT + torch.diag_embed(T(d))
    save("g26_lanczos_fp64", q_near=q, t_near=t, q_batch=qb, t_batch=tb, rrt_tv=root @ (root.mT @ T(tv)),
         iit_tv=inv @ (inv.mT @ T(tv)), A_tv=dense @ T(tv), Ainv_tv=np.linalg.solve(dense.numpy(), tv),
         checksum=cases.checksum(M, v0, Kd, V, C, d, v1, tv))
    print(f"  near {tuple(t.shape)}, batch {tuple(tb.shape)}, root {tuple(root.shape)}")


def g27_lanczos_fp32_fp64_divergence():
    """The g5 inputs (float32 data) through the reference's lanczos_tridiag in float32 AND in float64 (inputs promoted):
    the leading block on which the two runs agree is what a float32 implementation can be held to at 1e-4 -- beyond it
    the reference's own float32 run has left the exact recurrence (the g23 technique, for Lanczos)."""
    print("G27 lanczos_tridiag float32 vs float64 on identical float32 inputs")
    dev = torch.device("cpu")
    M = cases.spd_test_matrix(501, 100, dtype=np.float32, jitter=1e-6)
    v0 = cases.randn(502, 100, 1, dtype=np.float32)
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        q, t = lanczos_tridiag(T(M).to(dt).matmul, max_iter=100, dtype=dt, device=dev, matrix_shape=M.shape,
                               init_vecs=T(v0).to(dt))
        out[f"q_near_{tag}"], out[f"t_near_{tag}"] = q, t
    C, d, _ = cases.lowrank_diag(511, 2, 256, 8, 1)
    V = cases.randn(512, 2, 256, 3, dtype=np.float32)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C).to(dt)), DiagLinearOperator(T(d).to(dt)))
        q, t = lanczos_tridiag(A._matmul, max_iter=10, dtype=dt, device=dev, matrix_shape=A.matrix_shape,
                               batch_shape=A.batch_shape, init_vecs=T(V).to(dt))
        out[f"q_batch_{tag}"], out[f"t_batch_{tag}"] = q, t
    # cfg3-shaped: B = 2, N = 2048, R = 32, 4 probes, 20 steps
    C3, d3, _ = cases.lowrank_diag(2701, 2, 2048, 32, 1)
    V3 = cases.randn(2702, 2, 2048, 4, dtype=np.float32)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C3).to(dt)), DiagLinearOperator(T(d3).to(dt)))
        q, t = lanczos_tridiag(A._matmul, max_iter=20, dtype=dt, device=dev, matrix_shape=A.matrix_shape,
                               batch_shape=A.batch_shape, init_vecs=T(V3).to(dt))
        out[f"q_cfg3_{tag}"], out[f"t_cfg3_{tag}"] = q, t
    save("g27_lanczos_divergence", checksum=cases.checksum(M, v0, C, d, V, C3, d3, V3), **out)
    for k in ("near", "batch", "cfg3"):
        a, b = np.asarray(out[f"t_{k}_f32"], dtype=np.float64), np.asarray(out[f"t_{k}_f64"])
        n = min(a.shape[-1], b.shape[-1])
        print(f"  {k}: shapes {a.shape} / {b.shape}, max |t32 - t64| on the common block {np.abs(a[..., :n, :n] - b[..., :n, :n]).max():.3e}")


def g28_kronecker_roots():
    """KroneckerProductAddedDiagLinearOperator._root_decomposition / _root_inv_decomposition
    (kronecker_product_added_diag_linear_operator.py:224-294): the lazy Matmul roots of the constant-diagonal form and of
    the two Kronecker-structured-diagonal branches.  The roots themselves are unique only up to the signs of the
    eigenvectors, so the vectors hold R R^T, R_inv R_inv^T and the products R^T w / R_inv^T w squared (sign-free)."""
    import linear_operator
    from linear_operator.operators import KroneckerProductAddedDiagLinearOperator, KroneckerProductDiagLinearOperator

    print("G28 Kronecker roots")
    K1, K2, sig, _ = cases.kron_factors(2801, 2, 6, 8, 3)
    w = cases.randn(2802, 2, 48, 3, dtype=np.float32)
    d1 = (np.abs(cases.randn(2804, 2, 6, dtype=np.float32)) * 0.3 + 0.4).astype(np.float32)
    d2 = (np.abs(cases.randn(2805, 2, 8, dtype=np.float32)) * 0.3 + 0.4).astype(np.float32)
    c1 = np.array([[0.6], [0.9]], dtype=np.float32)
    c2 = np.array([[0.5], [0.3]], dtype=np.float32)
    out = {}
    with linear_operator.settings.max_cholesky_size(0):
        for tag in ("sigma", "full", "const"):
            Kp = KroneckerProductLinearOperator(DenseLinearOperator(T(K1)), DenseLinearOperator(T(K2)))
            if tag == "sigma":
                A = Kp + ConstantDiagLinearOperator(T(sig), 48)
            elif tag == "full":
                A = Kp + KroneckerProductDiagLinearOperator(DiagLinearOperator(T(d1)), DiagLinearOperator(T(d2)))
            else:
                # (the reference's constant-factor branch scales `evec_ * diag_values.sqrt()`, :247, and raises for batched
                #  constant factors -- [2, 6, 6] by [2, 1] -- so this branch is recorded for the first member, unbatched)
                Kp = KroneckerProductLinearOperator(DenseLinearOperator(T(K1[0])), DenseLinearOperator(T(K2[0])))
                A = Kp + KroneckerProductDiagLinearOperator(ConstantDiagLinearOperator(T(c1[0]), 6), ConstantDiagLinearOperator(T(c2[0]), 8))
            assert isinstance(A, KroneckerProductAddedDiagLinearOperator)
            wt = T(w[0]) if tag == "const" else T(w)
            R = A.root_decomposition().root
            Ri = A.root_inv_decomposition().root
            assert "Matmul" in type(R).__name__ and "Matmul" in type(Ri).__name__, (type(R), type(Ri))
            Rd, Rid = R.to_dense(), Ri.to_dense()
            out.update({f"{tag}_dense": A.to_dense(), f"{tag}_rrt": Rd @ Rd.mT, f"{tag}_riri": Rid @ Rid.mT,
                        f"{tag}_rtw_sq": (R._t_matmul(wt) ** 2).sum(-2), f"{tag}_ritw_sq": (Ri._t_matmul(wt) ** 2).sum(-2)})
    save("g28_kron_roots", checksum=cases.checksum(K1, K2, sig, w, d1, d2, c1, c2), **out)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["g28", "g27", "g26", "g25", "g24", "g23", "g22", "g21", "g20", "g19", "g18", "g17", "g16", "g15", "g14", "g13", "g12", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11"]
    for name, fn in (("g1", g1_linear_cg), ("g2", g2_pivoted_cholesky), ("g3", g3_preconditioner),
                     ("g4", g4_solve_and_inv_quad_logdet), ("g5", g5_lanczos), ("g6", g6_matmuls),
                     ("g7", g7_low_rank_root_added_diag), ("g8", g8_root_decomposition),
                     ("g9", g9_backward), ("g10", g10_backward_preconditioned), ("g11", g11_diagonalization),
                     ("g12", g12_kronecker_added_diag), ("g13", g13_minres),
                     ("g14", g14_sqrt_inv_matmul), ("g15", g15_lanczos_consumers_backward),
                     ("g16", g16_sum_operators), ("g17", g17_low_rank_root_added_diag_backward),
                     ("g18", g18_low_rank_root_added_diag_wide_root), ("g19", g19_kronecker_three_factors),
                     ("g20", g20_kronecker_structured_diag), ("g21", g21_minres_fp64),
                     ("g22", g22_kronecker_iteration_pinned), ("g23", g23_tridiag_divergence_and_tight_logdet),
                     ("g24", g24_kronecker_256_iteration_pinned), ("g25", g25_fp64_preconditioned_path),
                     ("g26", g26_lanczos_fp64), ("g27", g27_lanczos_fp32_fp64_divergence),
                     ("g28", g28_kronecker_roots)):
        if name in todo:
            fn()
    print("done")
