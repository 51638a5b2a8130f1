"""GPU tests of the drop-in boundary: the reference's own operator API (torch.linalg.solve / A.solve /
A.inv_quad_logdet / torch.logdet / pivoted_cholesky / linear_cg seam / preconditioner_override seam / probe
seam) running on the HIP path, checked against the golden vectors the real reference produced."""
import warnings
from unittest import mock

import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, max_rel_err_cols, rel_err

pytestmark = pytest.mark.gpu

import linear_operator_amd as lo  # noqa: E402
from linear_operator_amd import settings  # noqa: E402
from linear_operator_amd.operators import (  # noqa: E402
    AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator, DiagLinearOperator,
    KroneckerProductLinearOperator, LowRankRootLinearOperator, PsdSumLinearOperator, SumLinearOperator,
)
from linear_operator_amd.utils.warnings import NumericalWarning  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


class ProbedAddedDiag(AddedDiagLinearOperator):
    _probes = None

    def _probe_vectors_and_norms(self):  # hook: reference _linear_operator.py:629-633
        return self._probes


def test_solve_lowrank_operator_api_and_cg_seam():
    g = load_golden("g4_solve_lowrank")
    C, d, rhs = cases.lowrank_diag(401, 4, 2048, 32, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    spy = mock.MagicMock(wraps=lo.utils.linear_cg)
    with settings.cg_tolerance(1e-4), mock.patch("linear_operator_amd.utils.linear_cg", new=spy):
        x = torch.linalg.solve(A, dev(rhs))  # __torch_function__ -> solve -> Solve -> _solve -> utils.linear_cg
    assert spy.called and spy.call_args.kwargs["preconditioner"] is not None  # CG ran, preconditioned
    assert max_rel_err_cols(host(x), g["x"]) < 1e-4
    assert max_rel_err_cols(host(x), g["x_exact"]) < 1e-4
    # un-batched operator with a vector right-hand side (is_vector path, linear_cg.py:134-136,349-350)
    A0 = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C[0])), DiagLinearOperator(dev(d[0])))
    with settings.cg_tolerance(1e-4):
        x0 = A0.solve(dev(rhs[0, :, 0]))
    assert x0.shape == (2048,)
    assert rel_err(host(x0), g["x_exact"][0, :, 0]) < 1e-4


def test_inv_quad_logdet_probe_seam_and_logdet():
    g = load_golden("g4_iql_lowrank")
    C, d, rhs = cases.lowrank_diag(411, 3, 2048, 16, 1)
    Z, Zn = cases.probes(412, 3, 2048, 8)
    A = ProbedAddedDiag(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    A._probes = (dev(Z), dev(Zn))
    with settings.cg_tolerance(1e-4):
        iq, ld = A.inv_quad_logdet(dev(rhs), logdet=True)
    floor = 2048 * 1.2e-7 * 137.0
    assert np.allclose(host(iq), g["inv_quad"], rtol=1e-4)
    assert np.allclose(host(ld), g["logdet"], rtol=1e-4, atol=floor)
    # random probes (no hook): stochastic estimate, the reference's own acceptance is rtol 0.2 / atol 0.03 scale
    A2 = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    torch.manual_seed(0)
    with settings.cg_tolerance(1e-4), settings.num_trace_samples(64):
        ld2 = torch.logdet(A2)
    assert np.allclose(host(ld2), g["logdet_exact"], rtol=0.1, atol=2.0)


def test_inv_quad_logdet_deterministic_probes():
    """settings.deterministic_probes (reference functions/_inv_quad_logdet.py:80-105, settings.py:245-262): the base
    samples are drawn once and cached on the setting, coloured by a Lanczos root of the preconditioner; two calls give
    the SAME estimate, leaving the context drops the cache, the reference's DeprecationWarning is raised."""
    g = load_golden("g4_iql_lowrank")
    C, d, rhs = cases.lowrank_diag(411, 3, 2048, 16, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    assert settings.deterministic_probes.probe_vectors is None
    with settings.cg_tolerance(1e-4), settings.num_trace_samples(32), settings.deterministic_probes(True):
        with pytest.warns(DeprecationWarning, match="deterministic probes"):
            iq1, ld1 = A.inv_quad_logdet(dev(rhs), logdet=True)
        base = settings.deterministic_probes.probe_vectors
        assert base is not None and base.shape[-1] == 32 and base.shape[:-2] == (3,)
        with pytest.warns(DeprecationWarning):
            iq2, ld2 = A.inv_quad_logdet(dev(rhs), logdet=True)
        assert settings.deterministic_probes.probe_vectors is base
    assert settings.deterministic_probes.probe_vectors is None  # the cache lives as long as the context
    assert np.allclose(host(iq1), g["inv_quad"], rtol=1e-4)
    # same base samples; the Lanczos root of the preconditioner is recomputed per call from a random start vector (as
    # in the reference, whose precond_lt is rebuilt by the Function), so the two estimates agree to a few percent only
    assert np.allclose(host(ld1), host(ld2), rtol=0.1, atol=1.0)
    assert np.allclose(host(ld1), g["logdet_exact"], rtol=0.1, atol=2.0)


def test_kron_and_dense_operator_api():
    g = load_golden("g4_solve_kron")
    K1, K2, sig, rhs = cases.kron_factors(421, 2, 48, 48, 1)
    A = AddedDiagLinearOperator(KroneckerProductLinearOperator(dev(K1), dev(K2)),
                                ConstantDiagLinearOperator(dev(sig), 2304))
    with settings.cg_tolerance(1e-3):
        x = A.solve(dev(rhs))
    assert max_rel_err_cols(host(x), g["x"]) < 5e-3
    g = load_golden("g4_iql_dense")
    Kd, d, rhs = cases.dense_diag(431, 2, 2048, 1)
    Z, Zn = cases.probes(432, 2, 2048, 4)
    base = DenseLinearOperator(dev(Kd)).add_diagonal(dev(d))
    assert type(base) is AddedDiagLinearOperator
    A = ProbedAddedDiag(DenseLinearOperator(dev(Kd)), DiagLinearOperator(dev(d)))
    A._probes = (dev(Z), dev(Zn))
    with settings.cg_tolerance(1e-4):
        iq, ld = A.inv_quad_logdet(dev(rhs), logdet=True)
    assert np.allclose(host(iq), g["inv_quad"], rtol=1e-4)
    assert np.allclose(host(ld), g["logdet"], rtol=1e-4, atol=2048 * 1.2e-7 * 10.0)


def test_pivoted_cholesky_api_and_preconditioner_override():
    g = load_golden("g2_pivchol_lowrank")
    C = cases.lowrank_diag(242, 3, 2048, 32, 1)[0]
    L, piv = lo.pivoted_cholesky(LowRankRootLinearOperator(dev(C)), rank=15, return_pivots=True)
    assert np.array_equal(host(piv), g["piv_R32"]) and np.allclose(host(L), g["L_R32"], rtol=1e-4, atol=1e-5)
    m8 = cases.pivchol_dense8(201)
    g8 = load_golden("g2_pivchol_dense8")
    L8, p8 = lo.pivoted_cholesky(dev(m8), rank=3, return_pivots=True)  # plain tensor input
    assert np.array_equal(host(p8), g8["piv"]) and np.allclose(host(L8), g8["L"], rtol=1e-5, atol=1e-6)
    # preconditioner_override seam (reference added_diag_linear_operator.py:36-40,112-113)
    Cc, d, rhs = cases.lowrank_diag(401, 4, 2048, 32, 1)
    calls = []

    def override(self):
        calls.append(self)
        return (lambda t: t / self._diag_tensor._diag.unsqueeze(-1)), None, None  # Jacobi, opaque closure

    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(Cc)), DiagLinearOperator(dev(d)),
                                preconditioner_override=override)
    with settings.cg_tolerance(1e-4), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = A.solve(dev(rhs))
    assert calls
    ge = load_golden("g4_solve_lowrank")
    assert max_rel_err_cols(host(x), ge["x_exact"]) < 5e-3  # weaker preconditioner, still a solve


def test_linear_cg_signature_closures_warnings_errors():
    M = cases.spd_test_matrix(111, 10, dtype=np.float32)
    b = cases.randn(112, 10, 50, dtype=np.float32)
    Mt = dev(M)
    with warnings.catch_warnings(record=True) as ws:  # test_linear_cg.py:74-84: warning when tolerance unmet
        warnings.simplefilter("always")
        solves, t_mats = lo.utils.linear_cg(Mt.matmul, rhs=dev(b), n_tridiag=5, max_tridiag_iter=10, max_iter=10,
                                            tolerance=0, eps=1e-15)
    assert any(issubclass(w.category, NumericalWarning) for w in ws)
    assert tuple(t_mats.shape) == (5, 10, 10)
    actual = np.linalg.solve(M.astype(np.float64), b.astype(np.float64))
    assert np.allclose(host(solves), actual, atol=1e-3, rtol=1e-3)
    x_vec = lo.utils.linear_cg(lambda v: Mt @ v, dev(b[:, 0]), max_iter=10, max_tridiag_iter=5, tolerance=1e-6)
    assert x_vec.shape == (10,)
    Mn = M.copy()
    Mn[0, 0] = np.nan
    with pytest.raises(RuntimeError, match="NaNs encountered"):
        lo.utils.linear_cg(dev(Mn).matmul, dev(b), max_iter=10, max_tridiag_iter=5)
    # lanczos seam
    from linear_operator_amd.utils.lanczos import lanczos_tridiag, lanczos_tridiag_to_diag

    Ml = cases.spd_test_matrix(501, 100, dtype=np.float32, jitter=1e-6)
    q, t = lanczos_tridiag(dev(Ml).matmul, max_iter=100, dtype=torch.float32, device=Mt.device,
                           matrix_shape=Ml.shape, init_vecs=dev(cases.randn(502, 100, 1, dtype=np.float32)))
    assert np.allclose(host(q @ t @ q.mT), Ml, atol=1e-4)
    evals, evecs = lanczos_tridiag_to_diag(t[:20, :20].contiguous().unsqueeze(0))
    ref = np.linalg.eigvalsh(host(t[:20, :20]).astype(np.float64))
    assert np.allclose(host(evals)[0], np.where(ref >= 0, ref, 1.0), rtol=1e-4, atol=1e-5)


def test_ragged_shapes_multibatch_and_wide_rhs():
    """Odd sizes (N, R not multiples of the tiles), multi-dimensional batch, un-batched operator, 50 columns
    (the reference's own test width, test_linear_cg.py:50), all through the operator API with max_cholesky_size(0)."""
    from oracle import lo_oracle as orc

    rng = np.random.default_rng(99)
    # [2, 3] batch, N = 777, R = 7
    C = (rng.standard_normal((2, 3, 777, 7)) / np.sqrt(7)).astype(np.float32)
    d = (rng.random((2, 3, 777)) + 0.5).astype(np.float32)
    rhs = rng.standard_normal((2, 3, 777, 3)).astype(np.float32)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-5), settings.min_preconditioning_size(100):
        x = A.solve(dev(rhs))
        L, piv = A._linear_op.pivoted_cholesky(rank=5, return_pivots=True)
    dense = C.astype(np.float64) @ np.swapaxes(C.astype(np.float64), -1, -2)
    dense = dense + np.eye(777) * d.astype(np.float64)[..., None]
    exact = np.linalg.solve(dense, rhs.astype(np.float64))
    assert x.shape == (2, 3, 777, 3) and max_rel_err_cols(host(x), exact) < 1e-4
    Lo, pivo = orc.pivoted_cholesky(orc.LowRankRowSource(C), 5)
    assert tuple(L.shape) == (2, 3, 777, 5) and np.array_equal(host(piv), pivo) and np.array_equal(host(L), Lo)
    # un-batched operator, 50 columns, unpreconditioned (N < min_preconditioning_size)
    M = cases.spd_test_matrix(7, 100, dtype=np.float32)
    b = cases.randn(8, 100, 50, dtype=np.float32)
    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-6), settings.max_cg_iterations(100):
        xs = DenseLinearOperator(dev(M)).solve(dev(b))
    assert np.allclose(host(xs), np.linalg.solve(M.astype(np.float64), b.astype(np.float64)), atol=1e-3, rtol=1e-3)
    # rhs broadcast over the operator batch is rejected with the reference's message
    with settings.max_cholesky_size(0), pytest.raises(RuntimeError, match="same number of dimensions"):
        A.inv_quad_logdet(dev(rhs[0]), logdet=True)


def test_low_rank_root_added_diag_woodbury_closed_form():
    """`LowRankRoot + Diag` / `.add_diagonal` build LowRankRootAddedDiagLinearOperator (reference default routing);
    solve / logdet / inv_quad_logdet run the Woodbury closed form on the device (no CG) and match the reference."""
    from linear_operator_amd.operators import LowRankRootAddedDiagLinearOperator

    g = load_golden("g7_lowrank_added_diag")
    C, d, rhs = cases.lowrank_diag(701, 3, 1024, 16, 3)
    A = LowRankRootLinearOperator(dev(C)) + DiagLinearOperator(dev(d))
    assert isinstance(A, LowRankRootAddedDiagLinearOperator)
    spy = mock.MagicMock(wraps=lo.utils.linear_cg)
    with mock.patch("linear_operator_amd.utils.linear_cg", new=spy), settings.max_cholesky_size(0):
        x = A.solve(dev(rhs))
        ld = A.logdet()
        iq, ld2 = A.inv_quad_logdet(dev(rhs), logdet=True)
        iqn, none = A.inv_quad_logdet(dev(rhs), logdet=False, reduce_inv_quad=False)
    assert not spy.called and none is None
    assert max_rel_err_cols(host(x), g["x"]) < 1e-4 and max_rel_err_cols(host(x), g["x_exact"]) < 1e-5
    assert np.allclose(host(ld), g["logdet_exact"], rtol=1e-6) and np.allclose(host(ld2), g["logdet"], rtol=1e-5)
    assert np.allclose(host(iq), g["inv_quad"], rtol=1e-4) and np.allclose(host(iqn), g["inv_quad_noreduce"], rtol=1e-4)
    assert np.allclose(host(A.chol_cap_mat), g["chol_cap_mat"], rtol=1e-4, atol=1e-5)
    sig = np.array([[0.3], [0.7], [1.1]], dtype=np.float32)
    Ac = LowRankRootLinearOperator(dev(C)).add_diagonal(dev(sig))
    assert isinstance(Ac, LowRankRootAddedDiagLinearOperator)
    with settings.max_cholesky_size(0):
        assert max_rel_err_cols(host(Ac.solve(dev(rhs))), g["x_const"]) < 1e-4
        assert np.allclose(host(Ac.logdet()), g["logdet_const"], rtol=1e-5)
    # adding another diagonal keeps the class; anything else falls back to the CG operator (reference :105-115)
    assert isinstance(A + DiagLinearOperator(dev(d)), LowRankRootAddedDiagLinearOperator)
    assert type(A + DenseLinearOperator(torch.eye(1024, device="cuda"))) is AddedDiagLinearOperator
    # cfg3-sized batch: against the CG path of the explicit AddedDiagLinearOperator
    Cb, db, rb = cases.lowrank_diag(702, 8, 8192, 32, 1)
    Aw = LowRankRootLinearOperator(dev(Cb)) + DiagLinearOperator(dev(db))
    Acg = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(Cb)), DiagLinearOperator(dev(db)))
    with settings.cg_tolerance(1e-4):
        assert max_rel_err_cols(host(Aw.solve(dev(rb))), host(Acg.solve(dev(rb)))) < 1e-4


def test_root_decomposition_lanczos_consumers():
    """RootDecomposition.forward / root_decomposition / root_inv_decomposition /
    zero_mean_mvn_samples on the device (Lanczos + tridiagonal eigh + lo_root_from_lanczos_f32) against the golden
    outputs of the reference (sign-invariant products) and the oracle."""
    from linear_operator_amd.functions._root_decomposition import RootDecomposition
    from oracle import lo_oracle as orc

    g = load_golden("g8_root_decomposition")
    C, d, _ = cases.lowrank_diag(801, 2, 512, 8, 1)
    v1 = cases.randn(802, 2, 512, 1, dtype=np.float32)
    v3 = cases.randn(803, 2, 512, 3, dtype=np.float32)
    tv = cases.randn(804, 2, 512, 2, dtype=np.float32)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    for name, iv in (("p1", v1), ("p3", v3)):
        root, inv = RootDecomposition.apply(A.representation_tree(), 12, A.dtype, A.device, A.batch_shape,
                                            A.matrix_shape, True, True, dev(iv), *A.representation())
        assert tuple(root.shape) == g[f"root_{name}"].shape and tuple(inv.shape) == g[f"inv_{name}"].shape
        rrt = host(root @ (root.mT @ dev(tv)))
        iit = host(inv @ (inv.mT @ dev(tv)))
        assert max_rel_err_cols(rrt, g[f"rrt_tv_{name}"]) < 2e-3
        assert max_rel_err_cols(iit, g[f"iit_tv_{name}"]) < 1e-2
        ro, io = orc.root_decomposition(lambda v: orc.matvec_lowrank_diag(C, d, v), iv, 12)
        assert max_rel_err_cols(rrt, ro @ (np.swapaxes(ro, -1, -2) @ tv)) < 2e-3
    with settings.max_cholesky_size(0):
        Rinv = A.root_inv_decomposition(initial_vectors=dev(v3), test_vectors=dev(tv))
        R = A.root_decomposition()  # random start vector: only the quality of the approximation can be checked
        samples = A.zero_mean_mvn_samples(7)
    ri = Rinv.root.to_dense()
    assert max_rel_err_cols(host(ri @ (ri.mT @ dev(tv))), g["best_iit_tv"]) < 1e-2
    assert R.root.to_dense().shape[-2] == 512 and tuple(samples.shape) == (7, 2, 512)
    # Lanczos property (A q_0 in span(q_0, q_1)): R R^T reproduces A on the start vector up to the tridiagonal jitter
    root1, _ = RootDecomposition.apply(A.representation_tree(), 12, A.dtype, A.device, A.batch_shape, A.matrix_shape,
                                       True, False, dev(v1), *A.representation())
    assert max_rel_err_cols(host(root1 @ (root1.mT @ dev(v1))), host(A @ dev(v1))) < 5e-3
    # small operators take the dense Cholesky root (reference _choose_root_method)
    As = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C[:, :64])), DiagLinearOperator(dev(d[:, :64])))
    Rs = As.root_decomposition().to_dense()
    assert np.allclose(host(Rs), host(As.to_dense()), rtol=1e-4, atol=1e-5)


def test_diagonalization_lanczos_and_symeig():
    """SURVEY 8(f) rank 2, second half: `diagonalization(method="lanczos")` = Diagonalization.forward on the device
    Lanczos (start vector injected where the Function draws it) against golden g11 (sign-invariant quantities) and the
    oracle; complete 40 x 40 case and the symeig method against the true spectrum."""
    import linear_operator_amd as lo_pkg
    from oracle import lo_oracle as orc

    g = load_golden("g11_diagonalization")
    C, d, _ = cases.lowrank_diag(1201, 2, 384, 8, 1)
    tv = cases.randn(1203, 2, 384, 3, dtype=np.float32)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))

    def fake_randn(v):
        def f(*size, dtype=None, device=None, **kw):
            assert tuple(size) == v.shape
            return dev(v).to(dtype=dtype)
        return f

    with mock.patch("linear_operator_amd.utils.lanczos.torch.randn", side_effect=fake_randn(g["v0"])), \
            settings.max_root_decomposition_size(20):
        evals, evecs = A.diagonalization(method="lanczos")
    Q = evecs.to_dense()
    assert tuple(evals.shape) == g["evals"].shape and tuple(Q.shape) == g["evecs"].shape
    assert np.allclose(np.sort(host(evals), -1)[..., -8:], np.sort(g["evals"], -1)[..., -8:], rtol=2e-3)
    recon = host(Q @ (evals.unsqueeze(-1) * (Q.mT @ dev(tv))))
    assert max_rel_err_cols(recon, g["recon_tv"]) < 5e-3
    eo, qo = orc.diagonalization(lambda v: orc.matvec_lowrank_diag(C, d, v), np.broadcast_to(g["v0"], (2, 384, 1)).copy(), 20)
    assert max_rel_err_cols(recon, qo @ (eo[..., None] * (np.swapaxes(qo, -1, -2) @ tv))) < 5e-3
    # orthonormal columns
    assert np.abs(host(Q.mT @ Q) - np.eye(20, dtype=np.float32)).max() < 1e-3
    M = dev(g["dense_M"])
    with mock.patch("linear_operator_amd.utils.lanczos.torch.randn", side_effect=fake_randn(g["v1"])):
        e2, q2 = lo_pkg.diagonalization(M, method="lanczos")
    q2 = q2.to_dense()
    assert np.allclose(np.sort(host(e2)), np.sort(g["symeig_evals"]), rtol=1e-3, atol=1e-4)
    assert np.abs(host((q2 * e2) @ q2.mT) - g["dense_M"]).max() < 2e-3 * np.abs(g["dense_M"]).max()
    e3, q3 = lo_pkg.diagonalization(M)  # N <= max_cholesky_size -> symeig
    assert np.allclose(host(e3), g["symeig_evals"], rtol=1e-4, atol=1e-5)
    q3 = q3.to_dense()
    assert np.abs(host((q3 * e3) @ q3.mT) - g["dense_M"]).max() < 1e-4 * np.abs(g["dense_M"]).max()
    with pytest.raises(RuntimeError, match="Unknown diagonalization method"):
        A.diagonalization(method="qr")


def test_kronecker_added_diag_eig_closed_forms():
    """SURVEY 8(f) rank 3, second half: `KroneckerProduct + ConstantDiag` / `.add_diagonal(sigma2)` build the
    KroneckerProductAddedDiagLinearOperator of the reference's default routing: solve / logdet / inv_quad_logdet from
    the factors' eigendecompositions (no CG: the Kronecker matvec kernel applies Q and Q^T), values and gradients
    against golden g12 (the reference's own fp64 closed form and autograd)."""
    from linear_operator_amd.operators import KroneckerProductAddedDiagLinearOperator
    from linear_operator_amd import kernels as K

    g = load_golden("g12_kron_added_diag")
    K1, K2, _, rhs = cases.kron_factors(1301, 2, 24, 36, 3)
    sig = np.array([[0.3], [0.05]], dtype=np.float32)
    W = cases.randn(1302, 2, 864, 3, dtype=np.float32)

    def close(a, b, rel):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    def leaves():
        return [dev(x).clone().requires_grad_(True) for x in (K1, K2, sig, rhs)]

    k1, k2, st, rt = leaves()
    A = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(k2)) + ConstantDiagLinearOperator(st, 864)
    assert isinstance(A, KroneckerProductAddedDiagLinearOperator)
    K._hip.prof_enable(True)
    x = A.solve(rt)
    torch.cuda.synchronize()
    prof = K._hip.prof_report()
    K._hip.prof_enable(False)
    assert any(k.startswith("kron") for k in prof) and not any(k.startswith("cg") for k in prof), prof.keys()
    assert max_rel_err_cols(host(x), g["x_exact"]) < 1e-4 and max_rel_err_cols(host(x), g["x"]) < 1e-4
    (x * dev(W)).sum().backward()
    assert close(rt.grad, g["x_drhs"], 1e-3) and close(st.grad, g["x_dsig"], 2e-3)
    assert close(k1.grad, g["x_dK1"], 2e-3) and close(k2.grad, g["x_dK2"], 2e-3)
    k1, k2, st, rt = leaves()
    A = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(k2)).add_diagonal(st)
    assert isinstance(A, KroneckerProductAddedDiagLinearOperator)
    iq, ld = A.inv_quad_logdet(rt, logdet=True)
    assert np.allclose(host(iq), g["iq"], rtol=1e-4) and np.allclose(host(ld), g["ld_exact"], rtol=1e-5)
    (iq.sum() + (ld * dev(np.array([1.5, -0.5], dtype=np.float32))).sum()).backward()
    assert close(rt.grad, g["iql_drhs"], 1e-3) and close(st.grad, g["iql_dsig"], 2e-3)
    assert close(k1.grad, g["iql_dK1"], 2e-3) and close(k2.grad, g["iql_dK2"], 2e-3)
    assert np.allclose(host(A.logdet()), g["ld_exact"], rtol=1e-5)
    # a non-constant diagonal keeps the reference's CG branch (without a preconditioner, :132-134)
    dfull = dev(np.broadcast_to(sig, (2, 864)).copy())
    Ad = KroneckerProductLinearOperator(DenseLinearOperator(dev(K1)), DenseLinearOperator(dev(K2))) + DiagLinearOperator(dfull)
    assert isinstance(Ad, KroneckerProductAddedDiagLinearOperator) and not Ad._diag_is_constant
    with settings.cg_tolerance(1e-4), settings.max_cg_iterations(2000), warnings.catch_warnings():
        warnings.simplefilter("ignore", NumericalWarning)
        xd = Ad.solve(dev(rhs))
    assert max_rel_err_cols(host(xd), g["x_exact"]) < 5e-3


def test_backward_passes_against_reference_autograd():
    """gradients through Matmul / Solve / InvQuad / InvQuadLogdet on the HIP path (forward solves,
    the extra backward solve and the `_bilinear_derivative` contractions of csrc/lo_bilinear.hip) against the
    gradients the reference's autograd produced (golden g9; gradients are quadratic in CG solves stopped at 1e-5,
    hence max-norm comparisons at a few 1e-3)."""
    g = load_golden("g9_backward")
    C, d, rhs = cases.lowrank_diag(901, 2, 1024, 8, 3)
    W = cases.randn(902, 2, 1024, 3, dtype=np.float32)
    Z = cases.randn(903, 2, 1024, 6, dtype=np.float32)

    def close(a, b, rel=3e-3):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    def leaves(*arrs):
        return [dev(x).clone().requires_grad_(True) for x in arrs]

    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-5), settings.max_cg_iterations(200):
        Ct, dt, rt = leaves(C, d, rhs)
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        ((A @ rt) * dev(W)).sum().backward()
        assert close(Ct.grad, g["mm_dC"], 1e-5) and close(dt.grad, g["mm_dd"], 1e-5) and close(rt.grad, g["mm_drhs"], 1e-5)

        Ct, dt, rt = leaves(C, d, rhs)
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        x = A.solve(rt)
        (x * dev(W)).sum().backward()
        assert max_rel_err_cols(host(x), g["solve_x"]) < 1e-4
        assert close(Ct.grad, g["solve_dC"]) and close(dt.grad, g["solve_dd"]) and close(rt.grad, g["solve_drhs"])

        Ct, dt, rt = leaves(C, d, rhs)
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        iq = A.inv_quad(rt)
        iq.sum().backward()
        assert np.allclose(host(iq), g["iq"], rtol=1e-4)
        assert close(Ct.grad, g["iq_dC"]) and close(dt.grad, g["iq_dd"]) and close(rt.grad, g["iq_drhs"])

        Ct, dt, rt = leaves(C, d, rhs)
        A = ProbedAddedDiag(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        z = dev(Z)
        nrm = z.norm(dim=-2, keepdim=True)
        A._probes = (z / nrm, nrm)
        with settings.num_trace_samples(6):
            iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        assert np.allclose(host(iq), g["iql_iq"], rtol=1e-4) and np.allclose(host(ld), g["iql_ld"], rtol=1e-3, atol=1e-2)
        assert close(Ct.grad, g["iql_dC"]) and close(dt.grad, g["iql_dd"]) and close(rt.grad, g["iql_drhs"])

        Kd, _, rd = cases.dense_diag(904, 2, 300, 2)
        sig = np.array([[0.4], [0.9]], dtype=np.float32)
        Wd = cases.randn(905, 2, 300, 2, dtype=np.float32)
        Kt, st, rdt = leaves(Kd, sig, rd)
        Ad = AddedDiagLinearOperator(DenseLinearOperator(Kt), ConstantDiagLinearOperator(st, 300))
        xd = Ad.solve(rdt)
        (xd * dev(Wd)).sum().backward()
        assert max_rel_err_cols(host(xd), g["dense_x"]) < 1e-3
        assert close(Kt.grad, g["dense_dK"], 5e-3) and close(st.grad, g["dense_dsig"], 5e-3)
        assert close(rdt.grad, g["dense_drhs"], 5e-3)

        # Kronecker product + constant diagonal
        K1, K2, sk, rk = cases.kron_factors(907, 2, 12, 20, 2)
        Wk = cases.randn(908, 2, 240, 2, dtype=np.float32)
        with settings.max_cg_iterations(400):
            k1t, k2t, skt, rkt = leaves(K1, K2, sk, rk)
            Ak = AddedDiagLinearOperator(
                KroneckerProductLinearOperator(DenseLinearOperator(k1t), DenseLinearOperator(k2t)),
                ConstantDiagLinearOperator(skt, 240))
            ((Ak @ rkt) * dev(Wk)).sum().backward()
            assert close(k1t.grad, g["kron_mm_dK1"], 1e-5) and close(k2t.grad, g["kron_mm_dK2"], 1e-5)
            assert close(skt.grad, g["kron_mm_dsig"], 1e-4)
            k1t, k2t, skt, rkt = leaves(K1, K2, sk, rk)
            Ak = AddedDiagLinearOperator(
                KroneckerProductLinearOperator(DenseLinearOperator(k1t), DenseLinearOperator(k2t)),
                ConstantDiagLinearOperator(skt, 240))
            xk = Ak.solve(rkt)
            (xk * dev(Wk)).sum().backward()
            assert max_rel_err_cols(host(xk), g["kron_x"]) < 1e-3
            assert close(k1t.grad, g["kron_dK1"], 5e-3) and close(k2t.grad, g["kron_dK2"], 5e-3)
            assert close(skt.grad, g["kron_dsig"], 5e-3) and close(rkt.grad, g["kron_drhs"], 5e-3)

    # three Kronecker factors: regrouped into two dense groups, the group gradient pulled back to the factor
    K1, K2, s, vk = cases.kron_factors(906, 2, 4, 4, 1)
    k1 = dev(K1).requires_grad_(True)
    Ak3 = KroneckerProductLinearOperator(DenseLinearOperator(k1), DenseLinearOperator(dev(K2)),
                                         DenseLinearOperator(dev(K2)))
    v3 = cases.randn(909, 2, 64, 1, dtype=np.float32)
    (Ak3 @ dev(v3)).sum().backward()
    k64 = torch.from_numpy(K1).double().requires_grad_(True)
    k2d = torch.from_numpy(K2).double()
    dense3 = torch.stack([torch.kron(torch.kron(k64[b], k2d[b]), k2d[b]) for b in range(2)])
    (dense3 @ torch.from_numpy(v3).double()).sum().backward()
    assert np.abs(host(k1.grad) - k64.grad.numpy()).max() <= 1e-5 * np.abs(k64.grad.numpy()).max()


def test_pivoted_cholesky_vjp_hand_written_matches_autograd_tape():
    """The hand-written pull-back of the pivoted-Cholesky factor of a dense root (K = C C^T, R > m) against the
    generic differentiable re-expression (the reference's PivotedCholesky.backward construction)."""
    from linear_operator_amd import kernels as K
    from linear_operator_amd.functions._pivoted_cholesky import pivoted_cholesky_vjp
    C = cases.lowrank_diag(1101, 3, 2048, 32, 1)[0]
    G = cases.randn(1102, 3, 2048, 15, dtype=np.float32)
    op = LowRankRootLinearOperator(dev(C))
    L, perm = K.pivoted_cholesky(op._kernel_descriptor(), 15)
    (g_hand,) = pivoted_cholesky_vjp(op, perm, dev(G))
    (g_tape,) = pivoted_cholesky_vjp(op, perm, dev(G), generic=True)
    assert g_hand.shape == g_tape.shape == (3, 2048, 32)
    assert (g_hand - g_tape).abs().max().item() <= 2e-4 * g_tape.abs().max().item()


def test_backward_inv_quad_logdet_with_preconditioner_terms():
    """With the pivoted-Cholesky preconditioner the reference's gradient contains d logdet P and the probes' estimate of
    the same quantity, chained through PivotedCholesky.backward; here both are chained by hand
    (functions/_inv_quad_logdet._add_preconditioner_terms): same gradients for the same probes (golden g10)."""
    g = load_golden("g10_backward_precond")
    C, d, rhs = cases.lowrank_diag(1001, 2, 2048, 8, 1)
    Z = cases.randn(1002, 2, 2048, 6, dtype=np.float32)

    def close(a, b, rel=5e-3):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    z = dev(Z)
    nrm = z.norm(dim=-2, keepdim=True)
    with settings.cg_tolerance(1e-5), settings.max_cg_iterations(200), settings.num_trace_samples(6):
        Ct, dt, rt = [dev(x).clone().requires_grad_(True) for x in (C, d, rhs)]
        A = ProbedAddedDiag(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        A._probes = (z / nrm, nrm)
        iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        assert np.allclose(host(iq), g["iq"], rtol=1e-4) and np.allclose(host(ld), g["ld"], rtol=1e-3, atol=2e-2)
        assert close(rt.grad, g["drhs"]) and close(dt.grad, g["dd"]) and close(Ct.grad, g["dC"])
        Ct, dt, rt = [dev(x).clone().requires_grad_(True) for x in (C, d, rhs)]
        A = ProbedAddedDiag(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        A._probes = (z / nrm, nrm)
        ld = A.logdet()
        (ld * dev(np.array([1.5, -0.5], dtype=np.float32))).sum().backward()
        assert close(dt.grad, g["ld_dd"]) and close(Ct.grad, g["ld_dC"])
    # PivotedCholesky.backward through the public method
    Gl = cases.randn(1003, 2, 2048, 8, dtype=np.float32)
    Ct3 = dev(C).clone().requires_grad_(True)
    Lpc = LowRankRootLinearOperator(Ct3).pivoted_cholesky(rank=15)
    (Lpc * dev(Gl)).sum().backward()
    assert np.allclose(host(Lpc), g["pc_L"], rtol=1e-4, atol=1e-5) and close(Ct3.grad, g["pc_dC"], 1e-3)
    # constant diagonal
    sig = np.array([[0.6], [1.3]], dtype=np.float32)
    with settings.cg_tolerance(1e-5), settings.max_cg_iterations(200), settings.num_trace_samples(6):
        Ct2, st, rt2 = [dev(x).clone().requires_grad_(True) for x in (C, sig, rhs)]
        A = ProbedAddedDiag(LowRankRootLinearOperator(Ct2), ConstantDiagLinearOperator(st, 2048))
        A._probes = (z / nrm, nrm)
        iq, ld = A.inv_quad_logdet(rt2, logdet=True)
        (iq.sum() + ld.sum()).backward()
        assert np.allclose(host(iq), g["c_iq"], rtol=1e-4) and np.allclose(host(ld), g["c_ld"], rtol=1e-3, atol=2e-2)
        assert close(rt2.grad, g["c_drhs"]) and close(st.grad, g["c_dsig"]) and close(Ct2.grad, g["c_dC"])
    # and the stochastic gradient is close to the exact one (6 probes, preconditioned: a few percent)
    w = np.array([1.5, -0.5])[:, None]
    assert np.abs(host(dt.grad) - w * g["exact_dlogdet_dd"]).max() < 0.2 * np.abs(g["exact_dlogdet_dd"]).max()


def test_preconditioner_memo_hits_and_invalidates():
    """The preconditioner memo (operators/added_diag_linear_operator.py) serves rebuilt operators over the SAME tensors
    and is invalidated by in-place updates (tensor version counter) and by new tensors."""
    from linear_operator_amd import kernels as K
    from linear_operator_amd.operators import added_diag_linear_operator as adl

    C, d, rhs = cases.lowrank_diag(1101, 2, 2048, 8, 1)
    Ct, dt = dev(C), dev(d)
    K._hip.prof_enable(True)
    with settings.cg_tolerance(1e-4):
        x1 = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt)).solve(dev(rhs))
        torch.cuda.synchronize()
        first = K._hip.prof_report()
        x2 = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt)).solve(dev(rhs))
        torch.cuda.synchronize()
        second = K._hip.prof_report()
        dt.mul_(2.0)  # in place: same address, new version
        x3 = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt)).solve(dev(rhs))
        torch.cuda.synchronize()
        third = K._hip.prof_report()
    K._hip.prof_enable(False)
    built = lambda p: any(k.startswith("pc_") for k in p)  # noqa: E731
    assert built(first) and not built(second) and built(third)
    assert torch.equal(x1, x2)
    A3 = C.astype(np.float64) @ np.swapaxes(C.astype(np.float64), -1, -2) + np.stack([np.diag(2.0 * v) for v in d.astype(np.float64)])
    assert max_rel_err_cols(host(x3), np.linalg.solve(A3, rhs.astype(np.float64))) < 1e-4
    assert len(adl._precond_memo) <= adl.PRECONDITIONER_MEMO_SIZE


def test_minres_with_shifts_against_reference_and_oracle():
    """utils.minres (csrc/lo_minres.hip) -- several shifts, the (value = -1, per-member shifts) form
    of contour_integral_quad, a vector rhs, an all-zero column, the Woodbury preconditioner closure and an opaque
    Python closure -- against golden g13 (the reference), the oracle and the exact fp64 solves.  fp32 recurrences
    stopped at a 1e-4 relative update norm agree to a few 1e-4 (see tests/test_oracle_vs_golden.py)."""
    from linear_operator_amd.utils import minres
    from oracle import lo_oracle as orc

    g = load_golden("g13_minres")
    C, d, rhs = cases.lowrank_diag(1401, 2, 300, 8, 3)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    sh, sh2 = dev(g["sh"]), dev(g["sh2"])
    x = minres(A._matmul, dev(rhs), shifts=sh, max_iter=200)
    assert tuple(x.shape) == (3, 2, 300, 3)
    for q in range(3):
        assert max_rel_err_cols(host(x[q]), g["x_shifts"][q]) < 5e-4
        assert max_rel_err_cols(host(x[q]), g["x_exact"][q]) < 5e-4
    xo, info = orc.minres(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, shifts=g["sh"], max_iter=200)
    assert max_rel_err_cols(host(x).reshape(6, 300, 3), xo.reshape(6, 300, 3)) < 5e-4
    x2 = minres(A._matmul, dev(rhs), shifts=sh2, value=-1, max_iter=200)
    for q in range(3):
        assert max_rel_err_cols(host(x2[q]), g["x_ciq"][q]) < 5e-4
    A0 = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C[0])), DiagLinearOperator(dev(d[0])))
    x3 = minres(A0._matmul, dev(rhs[0, :, 0]), max_iter=200)
    assert tuple(x3.shape) == (300,) and np.abs(host(x3) - g["x_vec"]).max() < 5e-4 * np.abs(g["x_vec"]).max()
    rz = rhs.copy()
    rz[1, :, 2] = 0.0
    x4 = host(minres(A._matmul, dev(rz), shifts=sh[:2], max_iter=30))
    assert np.all(x4[:, 1, :, 2] == 0)
    keep = np.ones((2, 3), bool)
    keep[1, 2] = False
    for q in range(2):
        e = np.linalg.norm(x4[q] - g["x_zero_col"][q], axis=-2) / np.maximum(np.linalg.norm(g["x_zero_col"][q], axis=-2), 1e-30)
        assert e[keep].max() < 5e-4
    with settings.min_preconditioning_size(0), settings.max_preconditioner_size(4):
        pre, _, _ = A._preconditioner()
        x5 = minres(A._matmul, dev(rhs), shifts=sh, max_iter=200, preconditioner=pre)
    for q in range(3):
        assert max_rel_err_cols(host(x5[q]), g["x_precond"][q]) < 1e-3
    # opaque closures for the product and the preconditioner (called back once per iteration)
    dense = A.to_dense()
    x6 = minres(lambda v: dense @ v, dev(rhs), shifts=sh, max_iter=200, preconditioner=lambda v: v / dev(d).unsqueeze(-1))
    # (with a preconditioner P the shift enters the preconditioned recurrence: the systems solved are K + sigma P --
    # same in the reference; only sigma = 0 is comparable with the plain exact solve)
    assert max_rel_err_cols(host(x6[0]), g["x_exact"][0]) < 1e-3
    xo6, _ = orc.minres(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, shifts=g["sh"], max_iter=200,
                        preconditioner=lambda v: v / d[..., None])
    assert max_rel_err_cols(host(x6).reshape(6, 300, 3), xo6.reshape(6, 300, 3)) < 1e-3
    # a dense tensor as the "closure" (reference :36-37)
    x7 = minres(dense, dev(rhs), shifts=sh, max_iter=200)
    assert max_rel_err_cols(host(x7).reshape(6, 300, 3), g["x_exact"].reshape(6, 300, 3)) < 5e-4


def test_sqrt_inv_matmul_contour_integral_quadrature():
    """LinearOperator.sqrt_inv_matmul = contour integral quadrature over ONE shifted-MINRES run on
    the device: A^{-1/2} R, L A^{-1/2} R with the inverse quadratic form, A^{1/2} R through contour_integral_quad, and
    the backward pass -- against golden g14 (reference values and autograd gradients) and the exact matrix functions."""
    from linear_operator_amd.utils import contour_integral_quad

    g = load_golden("g14_sqrt_inv_matmul")
    C, d, rhs = cases.lowrank_diag(1501, 2, 300, 8, 3)
    lhs = cases.randn(1502, 2, 4, 300, dtype=np.float32)
    W = cases.randn(1503, 2, 300, 3, dtype=np.float32)
    W2 = cases.randn(1504, 2, 4, 3, dtype=np.float32)

    def close(a, b, rel):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    Ct, dt, rt = [dev(x).clone().requires_grad_(True) for x in (C, d, rhs)]
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
    res = A.sqrt_inv_matmul(rt)
    assert max_rel_err_cols(host(res), g["exact_inv_sqrt"]) < 5e-4 and max_rel_err_cols(host(res), g["res"]) < 5e-4
    (res * dev(W)).sum().backward()
    assert close(rt.grad, g["drhs"], 2e-3) and close(dt.grad, g["dd"], 5e-3) and close(Ct.grad, g["dC"], 5e-3)
    Ct, dt, rt, lt = [dev(x).clone().requires_grad_(True) for x in (C, d, rhs, lhs)]
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
    res2, iq = A.sqrt_inv_matmul(rt, lt)
    assert close(res2, g["l_res"], 1e-3) and np.allclose(host(iq), g["l_iq"], rtol=1e-3)
    ((res2 * dev(W2)).sum() + (iq * dev(np.array([[1.0, -0.5, 2.0, 0.3]], dtype=np.float32))).sum()).backward()
    assert close(rt.grad, g["l_drhs"], 2e-3) and close(lt.grad, g["l_dlhs"], 2e-3)
    assert close(dt.grad, g["l_dd"], 5e-3) and close(Ct.grad, g["l_dC"], 5e-3)
    with torch.no_grad():
        A0 = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
        solves, weights, _, shifts = contour_integral_quad(A0, dev(rhs), inverse=False)
    assert tuple(shifts.shape) == g["shifts"].shape and tuple(weights.shape) == g["weights"].shape
    assert np.allclose(host(shifts), g["shifts"], rtol=5e-2)
    sq = host((solves * weights).sum(0))
    assert max_rel_err_cols(sq, g["exact_sqrt"]) < 2e-4
    # settings.ciq_samples: zero_mean_mvn_samples = A^{1/2} z with the base samples drawn by torch.randn
    z = cases.randn(1505, 2, 300, 5, dtype=np.float32)
    with settings.ciq_samples(True), mock.patch("linear_operator_amd.operators._linear_operator.torch.randn",
                                                side_effect=lambda *a, **k: dev(z)):
        smp = A0.zero_mean_mvn_samples(5)
    assert tuple(smp.shape) == (5, 2, 300)
    A64 = (C.astype(np.float64) @ np.swapaxes(C.astype(np.float64), -1, -2)) + np.stack([np.diag(x) for x in d.astype(np.float64)])
    ev, Qe = np.linalg.eigh(A64)
    exact = (Qe * np.sqrt(ev)[..., None, :]) @ np.swapaxes(Qe, -1, -2) @ z.astype(np.float64)  # [2, 300, 5]
    assert max_rel_err_cols(np.moveaxis(host(smp), 0, -1), exact) < 5e-4
    # vector right-hand side, non-batch operator, functional form
    A1 = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C[0])), DiagLinearOperator(dev(d[0])))
    v = lo.sqrt_inv_matmul(A1, dev(rhs[0, :, 0]))
    assert tuple(v.shape) == (300,)
    assert np.abs(host(v) - g["exact_inv_sqrt"][0, :, 0]).max() < 5e-4 * np.abs(g["exact_inv_sqrt"][0, :, 0]).max()


def test_backward_of_root_decomposition_and_diagonalization():
    """RootDecomposition.backward / Diagonalization.backward (reference functions/_root_decomposition.py:104-171,
    _diagonalization.py:62-88): gradients of sign-invariant losses against golden g15 (the reference's autograd)."""
    from linear_operator_amd.functions._root_decomposition import RootDecomposition

    g = load_golden("g15_lanczos_consumers_backward")
    C, d, _ = cases.lowrank_diag(1601, 2, 256, 8, 1)
    v1 = cases.randn(1602, 2, 256, 1, dtype=np.float32)
    tv = cases.randn(1603, 2, 256, 2, dtype=np.float32)
    W1 = cases.randn(1604, 2, 256, 2, dtype=np.float32)
    W2 = cases.randn(1605, 2, 256, 2, dtype=np.float32)
    w = cases.randn(1608, 40, dtype=np.float32)
    sdiag = cases.randn(1609, 40, dtype=np.float32)
    Ws = cases.randn(1610, 40, 40, dtype=np.float32)

    def close(a, b, rel):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    for name, want_inv in (("both", True), ("root", False)):
        Ct, dt = dev(C).clone().requires_grad_(True), dev(d).clone().requires_grad_(True)
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Ct), DiagLinearOperator(dt))
        root, inv = RootDecomposition.apply(A.representation_tree(), 12, A.dtype, A.device, A.batch_shape,
                                            A.matrix_shape, True, want_inv, dev(v1), *A.representation())
        loss = ((root @ (root.mT @ dev(tv))) * dev(W1)).sum()
        if want_inv:
            loss = loss + ((inv @ (inv.mT @ dev(tv))) * dev(W2)).sum()
        loss.backward()
        assert abs(loss.item() - float(g[f"{name}_loss"])) < 2e-2 * abs(float(g[f"{name}_loss"]))
        assert close(Ct.grad, g[f"{name}_dC"], 3e-2) and close(dt.grad, g[f"{name}_dd"], 3e-2)
    M = dev(g["diag_M"]).clone().requires_grad_(True)
    with mock.patch("linear_operator_amd.utils.lanczos.torch.randn", side_effect=lambda *a, **k: dev(g["v2"])):
        evals, evecs = DenseLinearOperator(M).diagonalization(method="lanczos")
    q = evecs.to_dense()
    order = torch.argsort(evals)
    evs, qs = evals[order], q[:, order]
    loss = (evs * dev(w)).sum() + (((qs * dev(sdiag)) @ qs.mT) * dev(Ws)).sum()
    loss.backward()
    assert abs(loss.item() - float(g["diag_loss"])) < 2e-2 * abs(float(g["diag_loss"]))
    gm, rm = host(M.grad), g["diag_dM"]
    assert np.abs(0.5 * (gm + gm.T) - 0.5 * (rm + rm.T)).max() < 5e-2 * np.abs(rm).max()


# ------------------------------------------------------------------------------------------- multi-term sums (a6)
def _g16_inputs():
    C, d, rhs = cases.lowrank_diag(1601, 3, 2048, 16, 1)
    Kd, _, V = cases.dense_diag(1602, 3, 2048, 3)
    Kd = (Kd * np.float32(0.25)).astype(np.float32)
    Z, Zn = cases.probes(1603, 3, 2048, 6)
    wproj = cases.randn(1604, 2048, 2, dtype=np.float32)
    C2, _, _ = cases.lowrank_diag(1605, 3, 2048, 8, 1)
    return C, d, rhs, Kd, V, Z, Zn, wproj, C2


def test_sum_operators_lower_to_one_descriptor_and_match_the_reference():
    """Sum(LowRankRoot, Dense) + Diag and PsdSum(LowRankRoot, LowRankRoot) + Diag (reference
    sum_linear_operator.py:28-51, psd_sum_linear_operator.py:15-18): the whole tree lowers to ONE LO_OP_SUM descriptor
    -- matmul, the pivoted Cholesky of the sum (pivots bit-exact), solve and inv_quad_logdet run natively (no per-term
    Python matvec, no CG callback) and reproduce the reference's outputs (golden g16)."""
    from linear_operator_amd import kernels as K

    g = load_golden("g16_sum_operators")
    C, d, rhs, Kd, V, Z, Zn, wproj, C2 = _g16_inputs()
    S = SumLinearOperator(LowRankRootLinearOperator(dev(C)), DenseLinearOperator(dev(Kd)))
    A = ProbedAddedDiag(S, DiagLinearOperator(dev(d)))
    A._probes = (dev(Z), dev(Zn))
    desc = A._kernel_descriptor()
    assert desc is not None and desc.kind == K._hip.LO_OP_SUM and len(desc.terms) == 2 and desc.diag_mode == K._hip.LO_DIAG_FULL
    assert max_rel_err_cols(host(A._matmul(dev(V))), g["mv"]) < 1e-5
    assert max_rel_err_cols(host(S._matmul(dev(V))), g["mv_sum_only"]) < 1e-5
    L, piv = S.pivoted_cholesky(rank=15, return_pivots=True)
    m = g["pc_L"].shape[-1]
    assert L.shape[-1] == m and np.array_equal(host(piv)[..., :m], g["pc_piv"][..., :m])
    assert np.allclose(host(L), g["pc_L"], rtol=1e-5, atol=1e-6)
    K._hip.prof_enable(True)
    with settings.cg_tolerance(1e-4):
        x = A.solve(dev(rhs))
    torch.cuda.synchronize()
    prof = K._hip.prof_report()
    K._hip.prof_enable(False)
    assert "vec_axpy1" in prof and "dense_mv" in "".join(prof), "the sum must run through the lowered kernels"
    assert max_rel_err_cols(host(x), g["x"]) < 1e-4 and max_rel_err_cols(host(x), g["x_exact"]) < 1e-4
    spy = mock.MagicMock(wraps=lo.utils.linear_cg)
    with settings.cg_tolerance(1e-4), mock.patch("linear_operator_amd.utils.linear_cg", new=spy):
        iq, ld = A.inv_quad_logdet(dev(rhs), logdet=True)
    assert np.allclose(host(iq), g["iq"], rtol=1e-4)
    assert np.allclose(host(ld), g["ld"], rtol=1e-4, atol=2048 * 1.2e-7 * 150)
    # (6 probes: the stochastic estimate itself is ~10 % from the exact log-determinant, for the reference as for us)
    assert abs(float(host(ld).mean()) - float(g["logdet_exact"].mean())) < 0.2 * abs(float(g["logdet_exact"].mean()))
    # PsdSum of two roots
    P = PsdSumLinearOperator(LowRankRootLinearOperator(dev(C)), LowRankRootLinearOperator(dev(C2)))
    A2 = AddedDiagLinearOperator(P, DiagLinearOperator(dev(d)))
    assert A2._kernel_descriptor().kind == K._hip.LO_OP_SUM
    assert max_rel_err_cols(host(A2._matmul(dev(V))), g["psd_mv"]) < 1e-5
    L2, piv2 = P.pivoted_cholesky(rank=15, return_pivots=True)
    m2 = g["psd_pc_L"].shape[-1]
    assert L2.shape[-1] == m2 and np.array_equal(host(piv2)[..., :m2], g["psd_pc_piv"][..., :m2])
    assert np.allclose(host(L2), g["psd_pc_L"], rtol=1e-5, atol=1e-6)
    with settings.cg_tolerance(1e-4):
        x2 = A2.solve(dev(rhs))
    assert max_rel_err_cols(host(x2), g["psd_x"]) < 1e-4


def test_sum_operator_gradients_through_the_preconditioned_path():
    """inv_quad_logdet of Sum(LowRankRoot, Dense) + Diag with gradients: the logdet gradient chains through the pivoted
    Cholesky of the SUM (pivot columns through the differentiable Matmul, functions/_pivoted_cholesky.py) -- same
    gradients as the reference's autograd for the same probes (golden g16)."""
    g = load_golden("g16_sum_operators")
    C, d, rhs, Kd, V, Z, Zn, wproj, C2 = _g16_inputs()

    def close(a, b, rel=5e-3):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    with settings.cg_tolerance(1e-5), settings.max_cg_iterations(200):
        Ct, Kt, dt, rt = [dev(a).clone().requires_grad_(True) for a in (C, Kd, d, rhs)]
        A = ProbedAddedDiag(SumLinearOperator(LowRankRootLinearOperator(Ct), DenseLinearOperator(Kt)), DiagLinearOperator(dt))
        A._probes = (dev(Z), dev(Zn))
        iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
    assert np.allclose(host(iq), g["g_iq"], rtol=1e-4) and np.allclose(host(ld), g["g_ld"], rtol=1e-3, atol=2e-2)
    assert close(rt.grad, g["g_drhs"]) and close(dt.grad, g["g_dd"]) and close(Ct.grad, g["g_dC"])
    assert close(Kt.grad @ dev(wproj), g["g_dK_proj"]) and close(torch.diagonal(Kt.grad, dim1=-2, dim2=-1), g["g_dK_diag"])


def test_opaque_operator_pivoted_cholesky_through_the_row_fetch_callback():
    """An operator that does not lower to a descriptor: the pivoted Cholesky keeps the reference's generic row access
    (LinearOperator.__getitem__ with tensor indices, _linear_operator.py:2882-2902; here one `_t_matmul` on a one-hot
    column per pivot) as a callback of the same kernels -- nothing is densified, pivots and L equal the lowered path's
    -- and its AddedDiag solve / logdet gradient (diagonal) work on the generic path."""
    from linear_operator_amd.operators._linear_operator import LinearOperator

    class Opaque(LinearOperator):  # symmetric, only `_matmul` & co.
        def __init__(self, root):
            super().__init__(root)
            self.root = root

        def _matmul(self, rhs):
            return self.root @ (self.root.mT @ rhs)

        def _t_matmul(self, rhs):
            return self._matmul(rhs)

        def _size(self):
            return torch.Size((*self.root.shape[:-2], self.root.shape[-2], self.root.shape[-2]))

        def _transpose_nonbatch(self):
            return self

        def _diagonal(self):
            return (self.root ** 2).sum(-1)

        def _bilinear_derivative(self, left_vecs, right_vecs):
            return (left_vecs @ (right_vecs.mT @ self.root) + right_vecs @ (left_vecs.mT @ self.root),)

    C, d, rhs = cases.lowrank_diag(1701, 3, 2048, 12, 1)
    op = Opaque(dev(C))
    assert op._kernel_descriptor() is None
    with mock.patch.object(Opaque, "to_dense", side_effect=AssertionError("must not densify")):
        L, piv = op.pivoted_cholesky(rank=15, return_pivots=True)
    Lr, pivr = LowRankRootLinearOperator(dev(C)).pivoted_cholesky(rank=15, return_pivots=True)
    m = Lr.shape[-1]
    assert L.shape == Lr.shape and torch.equal(piv[..., :m], pivr[..., :m])
    assert np.allclose(host(L), host(Lr), rtol=2e-4, atol=2e-5)  # (rows through fp32 GEMMs instead of sequential dots)
    # element / row access of the generic class
    dense = torch.from_numpy(C) @ torch.from_numpy(C).mT
    assert np.allclose(host(op[1, 5, 7]), dense[1, 5, 7], rtol=1e-4)
    assert np.allclose(host(op[..., 3, :]), dense[..., 3, :], rtol=1e-4, atol=1e-5)
    # generic-path AddedDiag: solve and logdet gradients agree with the lowered operator's
    Z, Zn = cases.probes(1702, 3, 2048, 6)
    outs = []
    for make in (lambda c: Opaque(c), lambda c: LowRankRootLinearOperator(c)):
        Ct, dt = dev(C).clone().requires_grad_(True), dev(d).clone().requires_grad_(True)
        A = ProbedAddedDiag(make(Ct), DiagLinearOperator(dt))
        A._probes = (dev(Z), dev(Zn))
        with settings.cg_tolerance(1e-5), settings.max_cg_iterations(200):
            iq, ld = A.inv_quad_logdet(dev(rhs), logdet=True)
            (iq.sum() + ld.sum()).backward()
        outs.append((host(iq), host(ld), host(Ct.grad), host(dt.grad)))
    (iq0, ld0, dC0, dd0), (iq1, ld1, dC1, dd1) = outs
    assert np.allclose(iq0, iq1, rtol=1e-4) and np.allclose(ld0, ld1, rtol=1e-3, atol=2e-2)
    assert np.abs(dd0 - dd1).max() <= 5e-3 * np.abs(dd1).max() and np.abs(dC0 - dC1).max() <= 5e-3 * np.abs(dC1).max()


def test_low_rank_root_added_diag_gradients_match_reference_autograd():
    """`LowRankRoot + Diag` is what the default `+` routing builds: its inv_quad / logdet / solve must carry gradients
    to the root, the diagonal and the right-hand side (the reference differentiates the Woodbury expressions; here
    Solve / InvQuad Functions + the exact logdet gradient) -- golden g17 = the reference's autograd."""
    from linear_operator_amd.operators import LowRankRootAddedDiagLinearOperator

    g = load_golden("g17_lowrank_added_diag_backward")
    C, d, rhs = cases.lowrank_diag(1701, 3, 1024, 16, 3)
    W = cases.randn(1702, 3, 1024, 3, dtype=np.float32)

    def close(a, b, rel=2e-4):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    with settings.max_cholesky_size(0):
        Ct, dt, rt = [dev(a).clone().requires_grad_(True) for a in (C, d, rhs)]
        A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
        assert isinstance(A, LowRankRootAddedDiagLinearOperator)
        iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        assert np.allclose(host(iq), g["iq"], rtol=1e-4) and np.allclose(host(ld), g["ld"], rtol=1e-5)
        assert close(Ct.grad, g["dC"]) and close(dt.grad, g["dd"]) and close(rt.grad, g["drhs"])
        Ct, dt, rt = [dev(a).clone().requires_grad_(True) for a in (C, d, rhs)]
        A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
        x = A.solve(rt)
        (x * dev(W)).sum().backward()
        assert max_rel_err_cols(host(x), g["s_x"]) < 1e-4
        assert close(Ct.grad, g["s_dC"]) and close(dt.grad, g["s_dd"]) and close(rt.grad, g["s_drhs"])
        sig = np.array([[0.3], [0.7], [1.1]], dtype=np.float32)
        Ct, st = dev(C).clone().requires_grad_(True), dev(sig).clone().requires_grad_(True)
        Ac = LowRankRootLinearOperator(Ct).add_diagonal(st)
        ldc = Ac.logdet()
        (ldc * dev(np.array([1.0, -2.0, 0.5], dtype=np.float32))).sum().backward()
        assert np.allclose(host(ldc), g["c_ld"], rtol=1e-5) and close(Ct.grad, g["c_dC"]) and close(st.grad, g["c_dsig"])


def test_low_rank_root_added_diag_rank_above_32_matches_reference():
    """Roots wider than the kernels' 32-column Woodbury algebra take the library-GEMM route
    (low_rank_root_added_diag_linear_operator.py `_wide_root`): values and gradients against golden g18 (rank 48)."""
    from linear_operator_amd.operators import LowRankRootAddedDiagLinearOperator

    g = load_golden("g18_lowrank_added_diag_rank48")
    C, d, rhs = cases.lowrank_diag(1801, 2, 768, 48, 2)
    W = cases.randn(1802, 2, 768, 2, dtype=np.float32)
    assert cases.checksum(C, d, rhs, W) == g["checksum"]

    def close(a, b, rel=3e-4):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    with settings.max_cholesky_size(0):
        Ct, dt, rt = [dev(a).clone().requires_grad_(True) for a in (C, d, rhs)]
        A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
        assert isinstance(A, LowRankRootAddedDiagLinearOperator)
        iq, ld = A.inv_quad_logdet(rt, logdet=True)
        (iq.sum() + ld.sum()).backward()
        assert np.allclose(host(iq), g["iq"], rtol=1e-4) and np.allclose(host(ld), g["ld"], rtol=1e-5)
        assert close(Ct.grad, g["dC"]) and close(dt.grad, g["dd"]) and close(rt.grad, g["drhs"])
        Ct, dt, rt = [dev(a).clone().requires_grad_(True) for a in (C, d, rhs)]
        A = LowRankRootLinearOperator(Ct) + DiagLinearOperator(dt)
        x = A.solve(rt)
        (x * dev(W)).sum().backward()
        assert max_rel_err_cols(host(x), g["s_x"]) < 1e-4
        assert close(Ct.grad, g["s_dC"]) and close(dt.grad, g["s_dd"]) and close(rt.grad, g["s_drhs"])


def test_kronecker_product_of_three_factors_lowers_by_regrouping():
    """A Kronecker product of more than two dense factors is regrouped into two dense groups for the kernels
    (kronecker_product_linear_operator.py `_two_groups`); the gradient of a group is pulled back to its factors.
    Golden g19 = the reference's matmul / Cholesky-route solve / inv_quad_logdet and autograd gradients."""
    from linear_operator_amd.operators import KroneckerProductLinearOperator

    g = load_golden("g19_kron_three_factors")
    K1, K2, _, _ = cases.kron_factors(1901, 2, 6, 8, 3)
    K3, _, _, _ = cases.kron_factors(1902, 2, 10, 2, 1)
    rhs = cases.randn(1903, 2, 480, 3, dtype=np.float32)
    d = (np.abs(cases.randn(1904, 2, 480, dtype=np.float32)) * 0.2 + 0.3).astype(np.float32)
    W = cases.randn(1905, 2, 480, 3, dtype=np.float32)
    assert cases.checksum(K1, K2, K3, d, rhs, W) == g["checksum"]

    def close(a, b, rel):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    def build():
        lv = [dev(x).clone().requires_grad_(True) for x in (K1, K2, K3, d, rhs)]
        Kp = KroneckerProductLinearOperator(*(DenseLinearOperator(t) for t in lv[:3]))
        return lv, Kp, AddedDiagLinearOperator(Kp, DiagLinearOperator(lv[3]))

    lv, Kp, A = build()
    assert Kp._kernel_descriptor() is not None  # lowered, not the per-factor ATen chain
    assert max_rel_err_cols(host(Kp.matmul(lv[4])), g["mm_exact"]) < 1e-5
    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-5), settings.max_cg_iterations(2000), \
            settings.num_trace_samples(64):
        x = A.solve(lv[4])
        (x * dev(W)).sum().backward()
        assert max_rel_err_cols(host(x), g["x_exact"]) < 1e-3
        for t, name in zip(lv, ("x_dK1", "x_dK2", "x_dK3", "x_dd", "x_drhs")):
            assert close(t.grad, g[name], 5e-3), name
        lv, Kp, A = build()
        iq = A.inv_quad(lv[4])
        iq.sum().backward()
        assert np.allclose(host(iq), g["iq"], rtol=1e-3)
        # d iq / d K_i = -(x x^T contracted with the other factors): the iq part of g["iql_*"] isolated through the
        # dense inverse (the logdet part is stochastic on the CG route and is covered by the SLQ tests)
        for b in range(2):
            k1, k2, k3 = (k[b].astype(np.float64) for k in (K1, K2, K3))
            xb = g["x_exact"][b]
            Gt = (-(xb @ xb.T)).reshape(6, 8, 10, 6, 8, 10)
            assert close(lv[0].grad[b], np.einsum("iakjbl,ab,kl->ij", Gt, k2, k3), 5e-3)
            assert close(lv[1].grad[b], np.einsum("iakjbl,ij,kl->ab", Gt, k1, k3), 5e-3)
            assert close(lv[2].grad[b], np.einsum("iakjbl,ij,ab->kl", Gt, k1, k2), 5e-3)


def test_kronecker_product_plus_kronecker_structured_diagonal():
    """`KroneckerProduct + KroneckerProductDiag` (the multitask noise model): closed-form solve and logdet through the
    per-factor symmetrised eigendecomposition, gradients for the factors AND the diagonal factors (golden g20 = the
    reference's structured branches and autograd; its constant-factor logdet branch raises, so that logdet is checked
    against the dense fp64 value)."""
    from linear_operator_amd.operators import (ConstantDiagLinearOperator, KroneckerProductAddedDiagLinearOperator,
                                               KroneckerProductDiagLinearOperator, KroneckerProductLinearOperator)

    g = load_golden("g20_kron_structured_diag")
    K1, K2, _, _ = cases.kron_factors(2001, 2, 6, 8, 3)
    rhs = cases.randn(2002, 2, 48, 3, dtype=np.float32)
    W = cases.randn(2003, 2, 48, 3, dtype=np.float32)
    d1 = (np.abs(cases.randn(2004, 2, 6, dtype=np.float32)) * 0.3 + 0.4).astype(np.float32)
    d2 = (np.abs(cases.randn(2005, 2, 8, dtype=np.float32)) * 0.3 + 0.4).astype(np.float32)
    c1 = np.array([[0.6], [0.9]], dtype=np.float32)
    c2 = np.array([[0.5], [0.3]], dtype=np.float32)
    assert cases.checksum(K1, K2, rhs, W, d1, d2, c1, c2) == g["checksum"]
    wld = dev(np.array([1.5, -0.5], dtype=np.float32))

    def close(a, b, rel=2e-3):
        a, b = host(a), np.asarray(b)
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    with settings.max_cholesky_size(0):
        for tag, (a, b) in (("full", (d1, d2)), ("const", (c1, c2))):
            def build():
                lv = [dev(x).clone().requires_grad_(True) for x in (K1, K2, a, b, rhs)]
                if tag == "full":
                    D = KroneckerProductDiagLinearOperator(DiagLinearOperator(lv[2]), DiagLinearOperator(lv[3]))
                else:
                    D = KroneckerProductDiagLinearOperator(ConstantDiagLinearOperator(lv[2], 6),
                                                           ConstantDiagLinearOperator(lv[3], 8))
                A = KroneckerProductLinearOperator(DenseLinearOperator(lv[0]), DenseLinearOperator(lv[1])) + D
                assert isinstance(A, KroneckerProductAddedDiagLinearOperator)
                return lv, A

            lv, A = build()
            x = A.solve(lv[4])
            (x * dev(W)).sum().backward()
            assert max_rel_err_cols(host(x), g[f"{tag}_x_exact"]) < 1e-4
            for t, name in zip(lv, ("x_dK1", "x_dK2", "x_da", "x_db", "x_drhs")):
                assert close(t.grad, g[f"{tag}_{name}"]), (tag, name)
            lv, A = build()
            iq, ld = A.inv_quad_logdet(lv[4], logdet=True)
            assert np.allclose(host(iq), g[f"{tag}_iq"], rtol=1e-4)
            assert np.allclose(host(ld), g[f"{tag}_ld_exact"], rtol=1e-5)
            if tag == "full":
                (iq.sum() + (ld * wld).sum()).backward()
            else:
                iq.sum().backward()
            for t, name in zip(lv, ("iql_dK1", "iql_dK2", "iql_da", "iql_db", "iql_drhs")):
                assert close(t.grad, g[f"{tag}_{name}"]), (tag, name)


def test_host_api_with_max_preconditioner_size_above_32():
    """`settings.max_preconditioner_size(48)` through the operator API: solve and its gradients against the exact dense
    fp64 values / autograd, inv_quad_logdet forward + backward (the preconditioner-logdet correction and the
    pivoted-Cholesky pull-back run with a 48-column factor)."""
    Kd, d, rhs = cases.dense_diag(9300, 2, 1200, 3)
    W = cases.randn(9301, 2, 1200, 3, dtype=np.float32)
    Kt64 = torch.from_numpy(Kd).double().requires_grad_(True)
    dt64 = torch.from_numpy(d).double().requires_grad_(True)
    x64 = torch.linalg.solve(Kt64 + torch.diag_embed(dt64), torch.from_numpy(rhs).double())
    (x64 * torch.from_numpy(W).double()).sum().backward()
    ld64 = torch.logdet(Kt64.detach() + torch.diag_embed(dt64.detach()))

    def close(a, b, rel):
        a, b = host(a).astype(np.float64), b.detach().numpy()
        return a.shape == b.shape and np.abs(a - b).max() <= rel * np.abs(b).max()

    with settings.max_cholesky_size(0), settings.min_preconditioning_size(0), settings.max_preconditioner_size(48), \
            settings.cg_tolerance(1e-5), settings.num_trace_samples(64):
        Kt, dt = dev(Kd).clone().requires_grad_(True), dev(d).clone().requires_grad_(True)
        A = AddedDiagLinearOperator(DenseLinearOperator(Kt), DiagLinearOperator(dt))
        x = A.solve(dev(rhs))
        A._preconditioner()  # (the solve ran on the Function's rebuilt operator: build this instance's cache to look at it)
        assert A._woodbury.k == 48 and A._woodbury.Q.shape[-1] == 64
        (x * dev(W)).sum().backward()
        assert max_rel_err_cols(host(x), x64.detach().numpy()) < 1e-4
        sym = 0.5 * (Kt.grad + Kt.grad.mT)
        assert close(sym, 0.5 * (Kt64.grad + Kt64.grad.mT), 2e-3) and close(dt.grad, dt64.grad, 2e-3)
        Kt, dt = dev(Kd).clone().requires_grad_(True), dev(d).clone().requires_grad_(True)
        A = AddedDiagLinearOperator(DenseLinearOperator(Kt), DiagLinearOperator(dt))
        iq, ld = A.inv_quad_logdet(dev(rhs), logdet=True)
        (iq.sum() + ld.sum()).backward()
        assert np.allclose(host(ld), ld64.numpy(), rtol=2e-2)
        assert bool(torch.isfinite(Kt.grad).all()) and bool(torch.isfinite(dt.grad).all())


def test_preconditioner_rank_above_128_takes_the_reference_qr_route():
    """settings.max_preconditioner_size is unbounded in the reference (settings.py:417); the build kernels stop at rank
    128, beyond it the cache is built as the reference builds it (thin QR on the device, added_diag_linear_operator.py:
    161-184) and applied as a closure.  Solve and logdet against float64 dense algebra; fewer iterations than rank 15."""
    from linear_operator_amd import kernels as K

    N = 1500
    g = torch.Generator(device="cuda").manual_seed(150)
    x = torch.rand(N, 2, generator=g, device="cuda")
    Kd = torch.exp(-torch.cdist(x, x) ** 2 / (2 * 0.05 ** 2))  # (short length scale: the pivots do not run out before 150)
    Kd = ((Kd + Kd.mT) * 0.5).contiguous()
    d = torch.full((N,), 1e-2, device="cuda")
    rhs = torch.randn(N, 2, generator=g, device="cuda")
    want = torch.linalg.solve(Kd.double() + torch.diag(d.double()), rhs.double())
    counts = {}
    for rank in (15, 150):
        A = AddedDiagLinearOperator(DenseLinearOperator(Kd), DiagLinearOperator(d))
        with settings.max_preconditioner_size(rank), settings.min_preconditioning_size(100), settings.max_cholesky_size(0), \
                settings.cg_tolerance(1e-3), settings.max_cg_iterations(2000), settings.preconditioner_tolerance(1e-9):
            sol = A.solve(rhs)
            counts[rank] = K.cg_last_executed()["streaming_iterations"]
            closure, precond_lt, logdet_p = A._preconditioner()
        assert ((sol.double() - want).norm() / want.norm()).item() < 5e-3
        if rank == 150:
            assert closure.woodbury is None and A._q_cache.shape[-1] > 128
            L = A._piv_chol_self.double()
            ld_ref = torch.linalg.slogdet(L @ L.mT + torch.diag(d.double()))[1]
            assert abs(float(logdet_p) - float(ld_ref)) < 1e-2 * abs(float(ld_ref)) + 1e-2
    assert counts[150] < counts[15], counts


def test_nan_in_the_preconditioner_factor_warns_and_continues_without_it():
    """added_diag_linear_operator.py:126-131: NaNs in the pivoted-Cholesky factor (here: an operator with a negative
    diagonal, sqrt of the pivot) -> NumericalWarning and (None, None, None); a clean factor of the same shape does not
    warn.  (The test is one reduction pass here: amax propagates NaN.)"""
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo

    N = 300
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn(2, N, 20, generator=g, device="cuda")
    good = X @ X.mT
    bad = good.clone()
    bad[1] = -bad[1]  # second member: every diagonal entry negative
    d = torch.full((2, N), 0.5, device="cuda")
    with settings.min_preconditioning_size(10), settings.max_preconditioner_size(8):
        clear_preconditioner_memo()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = AddedDiagLinearOperator(DenseLinearOperator(bad), DiagLinearOperator(d))._preconditioner()
        assert out == (None, None, None)
        assert any(issubclass(x.category, NumericalWarning) and "NaNs encountered in preconditioner" in str(x.message)
                   for x in w)
        clear_preconditioner_memo()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            closure, lt, logdet = AddedDiagLinearOperator(DenseLinearOperator(good), DiagLinearOperator(d))._preconditioner()
        assert closure is not None and torch.isfinite(logdet).all()
        assert not any("NaNs encountered in preconditioner" in str(x.message) for x in w)


@pytest.mark.parametrize("batch,n", [((3,), 300), ((2, 2), 257), ((2,), 383)])
def test_batched_cholesky_of_257_to_383_rows_goes_through_the_padded_factorisation(batch, n):
    """utils/cholesky.py:_cholesky_ex: the batched float32 factorisation of these sizes kills the HIP context on this stack;
    psd_safe_cholesky (the N <= max_cholesky_size branch of solve / inv_quad_logdet, reference utils/cholesky.py:13-74)
    factorises blockdiag(A, I) instead.  Values against float64, and the small-operator solve through the API."""
    from linear_operator_amd.utils.cholesky import psd_safe_cholesky

    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn(*batch, n, 24, generator=g, device="cuda")
    A = X @ X.mT + 0.5 * torch.eye(n, device="cuda")
    L = psd_safe_cholesky(A)
    L64 = torch.linalg.cholesky(A.double())
    assert L.shape == A.shape and (L.double() - L64).abs().max().item() < 1e-4 * L64.abs().max().item()
    rhs = torch.randn(*batch, n, 3, generator=g, device="cuda")
    d = torch.full((*batch, n), 0.25, device="cuda")
    x = AddedDiagLinearOperator(DenseLinearOperator(A), DiagLinearOperator(d)).solve(rhs)
    exact = torch.linalg.solve(A.double() + torch.diag_embed(d.double()), rhs.double())
    assert ((x.double() - exact).norm(dim=-2) / exact.norm(dim=-2)).max().item() < 1e-4


@pytest.mark.parametrize("batch,n", [((3,), 600), ((2, 2), 513)])
def test_batched_cholesky_solve_of_one_column_above_512_rows(batch, n):
    """utils/cholesky.py:cholesky_solve: the batched single-column solve against factors of more than 512 rows kills the
    HIP context on this stack; small batched operators (N <= max_cholesky_size) with ONE right-hand-side column go through
    it on every `solve` (reference _linear_operator.py:2324-2379 -> cholesky()._cholesky_solve)."""
    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn(*batch, n, 24, generator=g, device="cuda")
    A = X @ X.mT
    d = torch.full((*batch, n), 0.5, device="cuda")
    rhs = torch.randn(*batch, n, 1, generator=g, device="cuda")
    x = AddedDiagLinearOperator(DenseLinearOperator(A), DiagLinearOperator(d)).solve(rhs)
    exact = torch.linalg.solve(A.double() + torch.diag_embed(d.double()), rhs.double())
    assert x.shape == rhs.shape
    assert ((x.double() - exact).norm(dim=-2) / exact.norm(dim=-2)).max().item() < 1e-4
