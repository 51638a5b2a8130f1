"""fp64 linear_cg (csrc/lo_cg_f64.hip) on the recipes of the reference's own test/utils/test_linear_cg.py:27-160 --
float64 operands, N = 100 / 10, vector / matrix / batched right-hand sides, initial guesses, CG-coefficient
tridiagonals -- against the golden vectors the REAL reference produced for exactly these inputs
(tests/golden/make_golden.py g1_linear_cg) and against the oracle.  Tolerances: 1e-9 relative on solutions, 1e-7 on the
tridiagonals (the reference's own acceptance is atol 1e-3 / rtol 1e-4)."""
import warnings

import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, rel_err
from oracle import lo_oracle as orc

pytestmark = pytest.mark.gpu

from linear_operator_amd.utils import linear_cg  # noqa: E402
from linear_operator_amd.utils.warnings import NumericalWarning  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


class Counting:
    def __init__(self, fn):
        self.fn, self.calls = fn, 0

    def __call__(self, v):
        self.calls += 1
        return self.fn(v)


def run(closure, rhs, **kw):
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        out = linear_cg(closure, rhs, **kw)
    return out, any(issubclass(w.category, NumericalWarning) for w in ws)


def test_fp64_n100_vector_matrix_and_initial_guess_match_the_reference():
    g = load_golden("g1_cg_fp64_n100")
    M = cases.spd_test_matrix(101, 100)
    b_vec, b_mat = cases.randn(102, 100), cases.randn(103, 100, 50)
    x0_vec, x0_mat = cases.randn(104, 100), cases.randn(105, 100, 50)
    Md = dev(M)
    runs = [(b_vec, None, "x_vec"), (b_vec, x0_vec, "x_vec_init"), (b_mat, None, "x_mat"), (b_mat, x0_mat, "x_mat_init")]
    for i, (b, x0, key) in enumerate(runs):
        kw = dict(max_iter=100)
        if x0 is not None:
            kw["initial_guess"] = dev(x0)
        x, warned = run(Md, dev(b), **kw)                      # dense tensor: the library's fp64 matvec
        assert x.dtype == torch.float64 and x.shape == g[key].shape
        assert rel_err(host(x), g[key]) < 1e-9, key
        assert warned == bool(g["warned"][i])
        cnt = Counting(Md.matmul)                              # closure: one callback per product, as the reference
        xc, _ = run(cnt, dev(b), **kw)
        assert cnt.calls == int(g["matvecs"][i]), key
        assert rel_err(host(xc), g[key]) < 1e-9, key
    # the reference test's own acceptance (test_linear_cg.py:47)
    x, _ = run(Md, dev(b_mat), max_iter=100)
    assert np.allclose(host(x), np.linalg.solve(M, b_mat), atol=1e-3, rtol=1e-4)


def test_fp64_tridiagonals_match_the_reference():
    g = load_golden("g1_cg_fp64_n10_tridiag")
    M = cases.spd_test_matrix(111, 10)
    b = cases.randn(112, 10, 50)
    cnt = Counting(dev(M).matmul)
    (x, t), warned = run(cnt, dev(b), n_tridiag=5, max_tridiag_iter=10, max_iter=10, tolerance=0, eps=1e-15)
    assert cnt.calls == int(g["matvecs"]) and warned == bool(g["warned"])
    assert t.shape == g["t_mat"].shape and t.dtype == torch.float64
    assert rel_err(host(x), g["x"]) < 1e-9 and rel_err(host(t), g["t_mat"]) < 1e-7
    eigs = np.linalg.eigvalsh(M)  # test_linear_cg.py:92-95
    for i in range(5):
        assert np.allclose(eigs, np.linalg.eigvalsh(host(t)[i]), atol=1e-3, rtol=1e-4)
    (x2, t2), _ = run(dev(M), dev(b), n_tridiag=5, max_tridiag_iter=10, max_iter=10, tolerance=0, eps=1e-15)
    assert rel_err(host(x2), g["x"]) < 1e-9 and rel_err(host(t2), g["t_mat"]) < 1e-7


def test_fp64_batched_and_batched_tridiagonals_match_the_reference():
    g = load_golden("g1_cg_fp64_batch")
    M = cases.spd_test_matrix(121, 100, batch=(5,))
    b = cases.randn(122, 5, 100, 50)
    cnt = Counting(dev(M).matmul)
    x, _ = run(cnt, dev(b), max_iter=100)
    assert cnt.calls == int(g["matvecs"]) and rel_err(host(x), g["x"]) < 1e-9
    x, _ = run(dev(M), dev(b), max_iter=100)
    assert rel_err(host(x), g["x"]) < 1e-9
    g = load_golden("g1_cg_fp64_batch_tridiag")
    M = cases.spd_test_matrix(131, 10, batch=(5,))
    b = cases.randn(132, 5, 10, 10)
    for closure in (dev(M), dev(M).matmul):
        (x, t), _ = run(closure, dev(b), n_tridiag=8, max_iter=10, max_tridiag_iter=10, tolerance=0, eps=1e-30)
        assert t.shape == g["t_mat"].shape
        assert rel_err(host(x), g["x"]) < 1e-9 and rel_err(host(t), g["t_mat"]) < 1e-7


def test_fp64_preconditioner_closure_zero_column_and_skip_against_the_oracle():
    """A Jacobi preconditioner closure, a zero right-hand-side column (masked, :178-181), the early exit when every
    column has converged at the initial guess (:207-208), an unbatched operator against a batched right-hand side."""
    M = cases.spd_test_matrix(141, 60)
    b = cases.randn(142, 3, 60, 4)
    b[1, :, 2] = 0.0
    dinv = 1.0 / np.diag(M)
    x_o, t_o, info = orc.linear_cg(lambda v: M @ v, b, n_tridiag=2, max_iter=60, max_tridiag_iter=12, tolerance=1e-8,
                                   preconditioner=lambda r: r * dinv[:, None])
    dd = dev(dinv)
    cnt = Counting(dev(M).matmul)
    (x, t), _ = run(cnt, dev(b), n_tridiag=2, max_iter=60, max_tridiag_iter=12, tolerance=1e-8,
                    preconditioner=lambda r: r * dd[:, None])
    assert cnt.calls == info.matvecs
    assert rel_err(host(x), x_o) < 1e-9 and t.shape == t_o.shape and rel_err(host(t), t_o) < 1e-7
    assert np.all(host(x)[1, :, 2] == 0.0)
    # exact initial guess: no iteration at all
    xs = np.linalg.solve(M, b)
    cnt = Counting(dev(M).matmul)
    x, _ = run(cnt, dev(b), initial_guess=dev(xs), max_iter=60)
    assert cnt.calls == 1 and rel_err(host(x), xs) < 1e-12


def test_fp64_operators_solve_and_differentiate_through_linear_cg():
    """fp64 operators reach the fp64 engine through the operator API (`_solve` -> utils.linear_cg with the bound
    `_matmul` as closure); the pull-backs of the solves run as library GEMMs.  Against torch autograd of the dense
    fp64 solve; tolerance 1e-3 relative: the reference's `eps = 1e-10` masks alpha once p^T A p < 1e-10, so its CG
    (and this one) stalls at a relative residual of ~1e-5 whatever the tolerance asks."""
    import linear_operator_amd as lo
    from linear_operator_amd.operators import (AddedDiagLinearOperator, DenseLinearOperator, DiagLinearOperator,
                                               LowRankRootLinearOperator)

    N = 200
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    X = torch.randn(N, N, generator=g, device="cuda", dtype=torch.float64)
    M0 = X @ X.T / N + torch.eye(N, device="cuda", dtype=torch.float64)
    C0 = torch.randn(N, 6, generator=g, device="cuda", dtype=torch.float64)
    d0 = torch.rand(N, generator=g, device="cuda", dtype=torch.float64) + 0.5
    b = torch.randn(N, 3, generator=g, device="cuda", dtype=torch.float64)

    def check(build, dense, leaves0):
        leaves = [t.clone().requires_grad_(True) for t in leaves0]
        with lo.settings.max_cholesky_size(0), lo.settings.cg_tolerance(1e-12), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            iq = build(*leaves).inv_quad(b)
            iq.sum().backward()
        ref_leaves = [t.clone().requires_grad_(True) for t in leaves0]
        ref = (b * torch.linalg.solve(dense(*ref_leaves), b)).sum()
        ref.backward()
        assert iq.dtype == torch.float64
        assert abs(iq.sum().item() - ref.item()) < 1e-6 * abs(ref.item())
        for a, r in zip(leaves, ref_leaves):
            assert a.grad.dtype == torch.float64
            assert (a.grad - r.grad).abs().max().item() < 1e-3 * r.grad.abs().max().item()

    check(lambda m: DenseLinearOperator(m), lambda m: m, [M0])
    check(lambda c, d: AddedDiagLinearOperator(LowRankRootLinearOperator(c), DiagLinearOperator(d)),
          lambda c, d: c @ c.T + torch.diag(d), [C0, d0])


def test_fp64_minres_matches_the_reference_recipes():
    """The reference's own test/utils/test_minres.py:17-80 (float64, value = -1, minres_tolerance 1e-6, shifts, batched
    and unbatched matrices and right-hand sides) on csrc/lo_minres_f64.hip, against the goldens the real reference
    produced (g21) -- dense tensors on the library's fp64 matvec and the same matrices as closures."""
    import linear_operator_amd as lo
    from linear_operator_amd.utils.minres import minres
    from test_oracle_vs_golden import minres64_inputs

    g = load_golden("g21_minres_fp64")
    for tag, M, b, sh in minres64_inputs():
        sh_t = None if sh is None else dev(sh)
        for closure in (dev(M), dev(M).matmul):
            with lo.settings.minres_tolerance(1e-6):
                x = minres(closure, rhs=dev(b), value=-1, shifts=sh_t)
            assert x.dtype == torch.float64 and tuple(x.shape) == g[f"x_{tag}"].shape, tag
            assert rel_err(host(x), g[f"x_{tag}"]) < 1e-9, tag


def test_float64_lanczos_and_root_decomposition_against_the_reference_golden():
    """Round 4: lanczos_tridiag in float64 (csrc/lo_lanczos_f64.hip) against the real reference's outputs (golden g26):
    dense tensor operator, batched with three start vectors, closure operator, and RootDecomposition's roots."""
    import cases
    from linear_operator_amd.functions._root_decomposition import RootDecomposition
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from linear_operator_amd.utils import lanczos

    g = load_golden("g26_lanczos_fp64")
    M = cases.spd_test_matrix(2601, 100, dtype=np.float64, jitter=1e-6)
    v0 = cases.randn(2602, 100, 1, dtype=np.float64)
    Md = dev(M)
    for closure in (Md, Md.matmul):  # tensor operand: the library's fp64 matvec; callable: called back per step
        q, t = lanczos.lanczos_tridiag(closure, max_iter=100, dtype=torch.float64, device=Md.device,
                                       matrix_shape=M.shape, init_vecs=dev(v0))
        assert q.dtype == torch.float64 and tuple(q.shape) == g["q_near"].shape and tuple(t.shape) == g["t_near"].shape
        q, t = host(q), host(t)
        assert np.allclose(t[:20, :20], g["t_near"][:20, :20], rtol=1e-8, atol=1e-12)
        assert np.allclose(q @ t @ q.T, M, atol=1e-9)  # test_lanczos.py:35-36 acceptance, at double's accuracy
        assert np.allclose(q.T @ q, np.eye(q.shape[1]), atol=1e-9)
    Kd, _, _ = cases.dense_diag(2603, 2, 300, 1, dtype=np.float64)
    V = cases.randn(2604, 2, 300, 3, dtype=np.float64)
    qb, tb = lanczos.lanczos_tridiag(dev(Kd), max_iter=16, dtype=torch.float64, device=Md.device,
                                     matrix_shape=Kd.shape[-2:], batch_shape=torch.Size([2]), init_vecs=dev(V))
    assert tuple(qb.shape) == g["q_batch"].shape
    assert np.allclose(host(tb), g["t_batch"], rtol=1e-9, atol=1e-12) and np.allclose(host(qb), g["q_batch"], atol=1e-9)
    C, d, _ = cases.lowrank_diag(2605, 2, 512, 8, 1, dtype=np.float64)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    v1 = cases.randn(2606, 2, 512, 1, dtype=np.float64)
    tv = dev(cases.randn(2607, 2, 512, 2, dtype=np.float64))
    root, inv = RootDecomposition.apply(A.representation_tree(), 14, A.dtype, A.device, A.batch_shape, A.matrix_shape,
                                        True, True, dev(v1), *A.representation())
    assert root.dtype == torch.float64
    assert rel_err(host(root @ (root.mT @ tv)), g["rrt_tv"]) < 1e-8
    assert rel_err(host(inv @ (inv.mT @ tv)), g["iit_tv"]) < 1e-8
    # and through the public entry (random start vector: accuracy against the dense matrix, not the golden)
    from linear_operator_amd.operators import DenseLinearOperator

    torch.manual_seed(2608)  # (the start vector is drawn inside: seeded, and the bar leaves room for its draw -- an
    #                          unseeded run of this line came out at 1.02e-8 once in about ten)
    with lo_settings().max_cholesky_size(0), lo_settings().max_root_decomposition_size(100):
        R = DenseLinearOperator(Md).root_decomposition().root.to_dense()
    assert R.dtype == torch.float64 and rel_err(host(R @ R.mT), M) < 5e-8


def lo_settings():
    import linear_operator_amd as lo

    return lo.settings


def test_fp64_is_limited_to_the_two_solvers_and_says_so():
    from linear_operator_amd import _hip, kernels as K

    C = torch.randn(2, 300, 4, device="cuda", dtype=torch.float64)
    with pytest.raises(_hip.HipExtensionError):
        K.lowrank_diag_descriptor(C, None)


def test_float64_preconditioned_path_against_the_reference_golden():
    """Round 4: float64 operators above `min_preconditioning_size` -- pivoted Cholesky on the float64 instantiation of the
    streaming kernels (pivots bit for bit against the real reference, golden g25), the QR preconditioner the reference
    builds (LAPACK on the device), `A.solve` and `inv_quad_logdet` with injected probes on lo_cg_solve_f64."""
    import cases
    import linear_operator_amd as lo
    from linear_operator_amd.operators import (AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator,
                                               DiagLinearOperator, LowRankRootLinearOperator)

    g = load_golden("g25_fp64_preconditioned")
    C, d, rhs = cases.lowrank_diag(2501, 2, 2048, 16, 2, dtype=np.float64)
    Z, Zn = cases.probes(2502, 2, 2048, 6, dtype=np.float64)
    Kd, _, _ = cases.dense_diag(2503, 2, 300, 1, dtype=np.float64)
    L, piv = LowRankRootLinearOperator(dev(C)).pivoted_cholesky(15, return_pivots=True)
    assert L.dtype == torch.float64 and np.array_equal(host(piv), g["piv_root"])
    assert np.allclose(host(L), g["L_root"], rtol=1e-9, atol=1e-11)
    Ld, pd_ = DenseLinearOperator(dev(Kd)).pivoted_cholesky(10, return_pivots=True)
    assert np.array_equal(host(pd_), g["piv_dense"]) and np.allclose(host(Ld), g["L_dense"], rtol=1e-9, atol=1e-11)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    with lo.settings.cg_tolerance(1e-8):
        x = A.solve(dev(rhs))
    assert x.dtype == torch.float64 and rel_err(host(x), g["x"]) < 1e-9 and rel_err(host(x), g["x_exact"]) < 1e-7

    class Probed(AddedDiagLinearOperator):
        _probes = None

        def _probe_vectors_and_norms(self):
            return self._probes

    Ap = Probed(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
    Ap._probes = (dev(Z), dev(Zn))
    with lo.settings.cg_tolerance(1e-8):
        iq, ld = Ap.inv_quad_logdet(dev(rhs), logdet=True)
    assert np.allclose(host(iq), g["inv_quad"], rtol=1e-9) and np.allclose(host(ld), g["logdet"], rtol=1e-7, atol=1e-7)
    Ac = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)),
                                 ConstantDiagLinearOperator(dev(np.full((2, 1), 0.7)), 2048))
    with lo.settings.cg_tolerance(1e-8):
        xc = Ac.solve(dev(rhs))
    assert rel_err(host(xc), g["x_const"]) < 1e-9
    # gradients through the float64 solve (library GEMM pull-backs)
    Cg, dg = dev(C).requires_grad_(True), dev(d).requires_grad_(True)
    Ag = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
    with lo.settings.cg_tolerance(1e-10):
        Ag.solve(dev(rhs)).sum().backward()
    dense = (Cg.detach() @ Cg.detach().mT + torch.diag_embed(dg.detach())).requires_grad_(True)
    torch.linalg.solve(dense, dev(rhs)).sum().backward()
    assert torch.allclose(dg.grad, dense.grad.diagonal(dim1=-1, dim2=-2), rtol=1e-6, atol=1e-9)


def test_float64_preconditioned_logdet_gradients_after_a_solve_on_the_same_tensors():
    """the dense (QR) preconditioner cache of float64 operators is built outside autograd and memoised under
    the operator's tensors; the derivative of logdet P and of the probes' term is chained by hand
    (functions/_inv_quad_logdet._add_preconditioner_terms).  With max_preconditioner_size = rank of the root the
    preconditioner IS the operator (P = A): logdet = logdet P and its gradient consists of the hand-chained terms only, no
    probe noise.  A solve fills the memo under no_grad FIRST; the gradients of inv_quad + logdet taken afterwards on the
    same tensors -- twice, the second time reusing the memo -- match dense autograd."""
    import cases
    import linear_operator_amd as lo
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo

    clear_preconditioner_memo()
    C, d, rhs = cases.lowrank_diag(2601, 2, 2048, 16, 1, dtype=np.float64)
    Cg, dg = dev(C).requires_grad_(True), dev(d).requires_grad_(True)
    y = dev(rhs)
    dense = (Cg.detach() @ Cg.detach().mT + torch.diag_embed(dg.detach())).requires_grad_(True)
    (torch.linalg.solve(dense, y).mul(y).sum() + torch.logdet(dense).sum()).backward()
    want_d = dense.grad.diagonal(dim1=-1, dim2=-2)
    want_C = (dense.grad + dense.grad.mT) @ Cg.detach()
    with lo.settings.cg_tolerance(1e-10), lo.settings.num_trace_samples(8), lo.settings.max_cg_iterations(200), \
            lo.settings.max_preconditioner_size(16), lo.settings.preconditioner_tolerance(1e-12):
        A0 = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
        with torch.no_grad():
            A0.solve(y)  # fills the preconditioner memo outside autograd
        for rep in range(2):
            torch.manual_seed(11 + rep)
            Cg.grad = dg.grad = None
            A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
            iq, ld = A.inv_quad_logdet(y, logdet=True)
            assert torch.allclose(ld, torch.logdet(dense.detach()), rtol=1e-8)
            (iq.sum() + ld.sum()).backward()  # (a graph kept in the memo would raise on the second pass)
            err_d = float((dg.grad - want_d).norm() / want_d.norm())
            err_C = float((Cg.grad - want_C).norm() / want_C.norm())
            assert err_d < 1e-6 and err_C < 1e-6, (rep, err_d, err_C)
    clear_preconditioner_memo()


@pytest.mark.parametrize("kind", ["constant_diag_operator", "diag_operator_with_equal_values"])
def test_float64_preconditioned_logdet_gradients_with_a_constant_diagonal(kind):
    """homoskedastic noise.  The dense closure of a constant diagonal is (t - q q^T t) / sigma with an
    UNSCALED q (added_diag_linear_operator.py:137-139), so diag(P^-1) = (1 - rowsum(q^2)) / sigma; with sigma != 1 the
    noise gradient of logdet P was wrong.  P = A here (preconditioner rank = root rank), so the gradient consists of the
    hand-chained terms only and has to match dense autograd."""
    import cases
    import linear_operator_amd as lo
    from linear_operator_amd.operators import (AddedDiagLinearOperator, ConstantDiagLinearOperator, DiagLinearOperator,
                                               LowRankRootLinearOperator)
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo

    clear_preconditioner_memo()
    B, N, R = 2, 2048, 16
    C, _, rhs = cases.lowrank_diag(2701, B, N, R, 1, dtype=np.float64)
    Cg = dev(C).requires_grad_(True)
    y = dev(rhs)
    sig = torch.tensor([[0.37], [2.5]], dtype=torch.float64, device="cuda", requires_grad=True)  # [B, 1], far from 1
    dense = (Cg.detach() @ Cg.detach().mT + torch.diag_embed(sig.detach().expand(B, N))).requires_grad_(True)
    (torch.linalg.solve(dense, y).mul(y).sum() + torch.logdet(dense).sum()).backward()
    want_sig = dense.grad.diagonal(dim1=-1, dim2=-2).sum(-1, keepdim=True)
    want_C = (dense.grad + dense.grad.mT) @ Cg.detach()
    with lo.settings.cg_tolerance(1e-10), lo.settings.num_trace_samples(8), lo.settings.max_cg_iterations(200), \
            lo.settings.max_preconditioner_size(R), lo.settings.preconditioner_tolerance(1e-12):
        for rep in range(2):
            torch.manual_seed(21 + rep)
            Cg.grad = sig.grad = None
            if kind == "constant_diag_operator":
                D = ConstantDiagLinearOperator(sig, diag_shape=N)
            else:
                D = DiagLinearOperator(sig.expand(B, N))
            A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), D)
            iq, ld = A.inv_quad_logdet(y, logdet=True)
            assert torch.allclose(ld, torch.logdet(dense.detach()), rtol=1e-8)
            (iq.sum() + ld.sum()).backward()
            err_s = float((sig.grad - want_sig).norm() / want_sig.norm())
            err_C = float((Cg.grad - want_C).norm() / want_C.norm())
            assert err_s < 1e-6 and err_C < 1e-6, (kind, rep, err_s, err_C)
    clear_preconditioner_memo()
