"""CPU tests of the host-side mirror: settings, operator algebra / dispatch, representation tree, the
Cholesky plumbing branch (cfg1), error conventions, the C-ABI library (loads + exports every declared symbol)
and the 'no CPU fallback' rule.  No GPU compute here."""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest
import torch

import cases
from conftest import ROOT, load_golden

import linear_operator_amd as lo
from linear_operator_amd import _hip, settings
from linear_operator_amd.operators import (
    AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator, DiagLinearOperator,
    KroneckerProductLinearOperator, LinearOperator, LowRankRootLinearOperator, PsdSumLinearOperator,
    RootLinearOperator, SumLinearOperator,
)

T = torch.from_numpy


def test_settings_names_defaults_and_context_protocol():
    # names and defaults GPyTorch relies on
    assert settings.cg_tolerance.value() == 1
    assert settings.max_cg_iterations.value() == 1000
    assert settings.max_cholesky_size.value() == 800
    assert settings.max_lanczos_quadrature_iterations.value() == 20
    assert settings.max_preconditioner_size.value() == 15
    assert settings.min_preconditioning_size.value() == 2000
    assert settings.num_trace_samples.value() == 10
    assert settings.preconditioner_tolerance.value() == 1e-3
    assert settings.terminate_cg_by_size.off() and settings.skip_logdet_forward.off() and settings.verbose_linalg.off()
    assert settings.fast_computations.solves.on() and settings.fast_computations.log_prob.on()
    with settings.cg_tolerance(1e-4), settings.max_cholesky_size(0), settings.fast_computations(solves=False):
        assert settings.cg_tolerance.value() == 1e-4 and settings.max_cholesky_size.value() == 0
        assert settings.fast_computations.solves.off() and settings.fast_computations.log_prob.on()
    assert settings.cg_tolerance.value() == 1 and settings.fast_computations.solves.on()
    with settings.cholesky_jitter(float_value=1e-3):
        assert settings.cholesky_jitter.value(torch.float) == 1e-3
    assert settings.cholesky_jitter.value(torch.float) == 1e-6

    class my_tol(settings.cg_tolerance):  # GPyTorch-style subclassing keeps working
        pass

    with my_tol(0.5):
        assert my_tol.value() == 0.5


def test_abi_library_loads_and_exports_every_declared_symbol():
    lib = _hip.load()
    assert lib.lo_abi_version() == _hip.ABI_VERSION and lib.lo_target_arch() == b"gfx950"
    hdr = open(os.path.join(ROOT, "include", "lo_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(lo_[a-z0-9_]+)\s*\(", hdr)) - {"lo_matvec_cb"})
    assert len(declared) >= 18
    raw = ctypes.CDLL(_hip.lib_path())
    for name in declared:
        assert hasattr(raw, name), f"liblo_amd.so does not export {name} declared in include/lo_amd.h"
    assert set(_hip.EXPORTS) == set(declared)


def test_class_relationships_and_add_routing():
    # isinstance relations are part of the contract
    assert issubclass(AddedDiagLinearOperator, SumLinearOperator)
    assert issubclass(LowRankRootLinearOperator, RootLinearOperator)
    assert issubclass(ConstantDiagLinearOperator, DiagLinearOperator)
    assert issubclass(PsdSumLinearOperator, SumLinearOperator)
    C = torch.randn(3, 20, 4)
    d = torch.rand(3, 20) + 0.5
    A = RootLinearOperator(C) + DiagLinearOperator(d)
    assert type(A) is AddedDiagLinearOperator and A._linear_op.__class__ is RootLinearOperator
    A2 = DenseLinearOperator(torch.randn(3, 20, 20)).add_diagonal(d)
    assert type(A2) is AddedDiagLinearOperator and isinstance(A2._diag_tensor, DiagLinearOperator)
    A3 = DenseLinearOperator(torch.randn(20, 20)).add_jitter(1e-2)
    assert isinstance(A3._diag_tensor, ConstantDiagLinearOperator)
    assert type(A + DiagLinearOperator(d)) is AddedDiagLinearOperator
    S = RootLinearOperator(C) + DenseLinearOperator(torch.randn(3, 20, 20))
    assert type(S) is SumLinearOperator and type(S + DiagLinearOperator(d)) is AddedDiagLinearOperator
    with pytest.raises(RuntimeError, match="only have two components"):
        AddedDiagLinearOperator(RootLinearOperator(C), DiagLinearOperator(d), DiagLinearOperator(d))
    with pytest.raises(RuntimeError, match="must be a DiagLinearOperator"):
        AddedDiagLinearOperator(RootLinearOperator(C), RootLinearOperator(C))
    with pytest.raises(ValueError, match="final singleton dimension"):
        ConstantDiagLinearOperator(torch.rand(3, 20), 20)


def test_matmul_to_dense_and_torch_function_dispatch_cpu():
    g = load_golden("g6_matmul")
    C, d, v = cases.lowrank_diag(601, 3, 256, 8, 5)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    y = torch.matmul(A, T(v))  # __torch_function__ -> matmul -> Matmul Function -> _matmul (ATen on CPU)
    assert np.allclose(y.numpy(), g["y_lowrank_diag"], rtol=1e-5, atol=1e-5)
    assert np.allclose((A @ T(v)).numpy(), g["y_lowrank_diag"], rtol=1e-5, atol=1e-5)
    assert np.allclose(A.to_dense().numpy(), (C @ np.swapaxes(C, -1, -2)) + np.stack([np.diag(x) for x in d]),
                       rtol=1e-4, atol=1e-5)
    K1, K2, s, vk = cases.kron_factors(621, 2, 12, 20, 3)
    kp = KroneckerProductLinearOperator(T(K1), T(K2))
    assert np.allclose(kp._matmul(T(vk)).numpy(), g["y_kron"], rtol=1e-4, atol=1e-4)
    assert np.allclose(kp._diagonal().numpy(), g["diag_kron"], rtol=1e-6)
    assert np.allclose(kp.to_dense().numpy(), np.stack([np.kron(K1[i], K2[i]) for i in range(2)]), rtol=1e-5)
    Ak = AddedDiagLinearOperator(kp, ConstantDiagLinearOperator(T(s), 240))
    assert np.allclose(Ak._matmul(T(vk)).numpy(), g["y_kron_diag"], rtol=1e-4, atol=1e-4)
    with pytest.raises(NotImplementedError, match="is not implemented"):
        torch.trace(A)
    with pytest.raises(RuntimeError):
        torch.matmul(A, torch.randn(3, 255, 2))


def test_representation_tree_roundtrip():
    C, d, _ = cases.lowrank_diag(5, 2, 30, 4, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), ConstantDiagLinearOperator(torch.rand(2, 1), 30))
    rep = A.representation()
    assert len(rep) == 2 and all(torch.is_tensor(t) for t in rep)
    B = A.representation_tree()(*rep)
    assert type(B) is AddedDiagLinearOperator and type(B._linear_op) is LowRankRootLinearOperator
    assert type(B._diag_tensor) is ConstantDiagLinearOperator and B._diag_tensor.diag_shape == 30
    assert torch.equal(B.to_dense(), A.to_dense())


def test_cfg1_dense256_cholesky_branch_matches_reference():
    """BASELINE cfg1: torch.linalg.solve(DenseLinearOperator(K256), b) takes the Cholesky branch (no CG, no GPU)."""
    g = load_golden("g4_cfg1_dense256")
    M = cases.spd_test_matrix(441, 256, dtype=np.float32, jitter=1.0)
    b = cases.randn(442, 256, 3, dtype=np.float32)
    called = []
    orig = lo.utils.linear_cg
    lo.utils.linear_cg = lambda *a, **k: called.append(1) or orig(*a, **k)
    try:
        x = torch.linalg.solve(DenseLinearOperator(T(M)), T(b))
    finally:
        lo.utils.linear_cg = orig
    assert not called
    assert np.allclose(x.numpy(), g["x"], rtol=1e-4, atol=1e-5)
    iq, ld = DenseLinearOperator(T(M)).inv_quad_logdet(T(b), logdet=True)
    assert np.allclose(ld.item(), np.linalg.slogdet(M.astype(np.float64))[1], rtol=1e-4)
    assert np.allclose(iq.item(), (b * np.linalg.solve(M.astype(np.float64), b)).sum(), rtol=1e-4)


def test_no_cpu_fallback_for_the_iterative_path():
    C, d, rhs = cases.lowrank_diag(6, 2, 64, 4, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(T(C)), DiagLinearOperator(T(d)))
    with settings.max_cholesky_size(0):
        with pytest.raises(_hip.HipExtensionError, match="no CPU fallback"):
            A.solve(T(rhs))
        with pytest.raises(_hip.HipExtensionError):
            A.inv_quad_logdet(T(rhs), logdet=True)
    with pytest.raises(_hip.HipExtensionError):
        lo.utils.linear_cg(T(np.eye(8, dtype=np.float32)).matmul, torch.randn(8, 1), max_iter=8, max_tridiag_iter=4)
    assert A._kernel_descriptor() is None  # CPU tensors never lower to a kernel descriptor


def test_linear_cg_argument_errors_match_reference():
    M = torch.eye(10)
    with pytest.raises(RuntimeError, match="larger than the number of CG iterations"):  # linear_cg.py:159-160
        lo.utils.linear_cg(M.matmul, torch.randn(10), max_iter=5)
    with pytest.raises(RuntimeError, match="must be a tensor, or a callable"):  # :163-166
        lo.utils.linear_cg(3.0, torch.randn(10), max_iter=10, max_tridiag_iter=5)
    A = DenseLinearOperator(torch.randn(3, 20, 21))
    with pytest.raises(RuntimeError, match="square"):
        A.solve(torch.randn(3, 21, 1))
    sq = DenseLinearOperator(torch.eye(900))
    with pytest.raises(RuntimeError, match="same number of dimensions"):
        sq.inv_quad_logdet(torch.randn(2, 900, 1), logdet=True)
    with pytest.raises(RuntimeError, match="must be specifed"):
        sq.inv_quad_logdet(None, logdet=False)


def test_kronecker_added_diag_routing_and_closed_form_plumbing_cpu():
    """`KroneckerProduct + ConstantDiag / Diag / .add_diagonal` build the KroneckerProductAddedDiagLinearOperator like the
    reference (kronecker_product_linear_operator.py:98-145); its logdet (pure ATen eigh plumbing) against golden g12 and
    the symeig diagonalization of a Kronecker product are checked here, the solves (Kronecker matvec kernel) on the GPU."""
    from linear_operator_amd.operators import KroneckerProductAddedDiagLinearOperator

    torch.set_num_threads(1)  # (torch's fp64 LAPACK calls hang with many threads in the build container)
    g = load_golden("g12_kron_added_diag")
    K1, K2, _, rhs = cases.kron_factors(1301, 2, 24, 36, 3)
    sig = np.array([[0.3], [0.05]], dtype=np.float32)
    kp = KroneckerProductLinearOperator(DenseLinearOperator(T(K1)), DenseLinearOperator(T(K2)))
    a1 = kp + ConstantDiagLinearOperator(T(sig), 864)
    a2 = kp.add_diagonal(T(sig))
    a3 = kp.add_diagonal(torch.tensor(0.3))
    a4 = kp + DiagLinearOperator(T(np.broadcast_to(sig, (2, 864)).copy()))
    for a, const in ((a1, True), (a2, True), (a3, True), (a4, False)):
        assert isinstance(a, KroneckerProductAddedDiagLinearOperator) and isinstance(a, AddedDiagLinearOperator)
        assert a._diag_is_constant == const and a._preconditioner() == (None, None, None)
    assert isinstance(AddedDiagLinearOperator(kp, ConstantDiagLinearOperator(T(sig), 864)), AddedDiagLinearOperator)
    assert not isinstance(AddedDiagLinearOperator(kp, ConstantDiagLinearOperator(T(sig), 864)),
                          KroneckerProductAddedDiagLinearOperator)  # the explicit class keeps the CG path
    assert isinstance(a1 + ConstantDiagLinearOperator(T(sig), 864), KroneckerProductAddedDiagLinearOperator)
    assert np.allclose(a1.logdet().numpy(), g["ld_exact"], rtol=1e-5)
    _, ld = a2.inv_quad_logdet(None, logdet=True)
    assert np.allclose(ld.numpy(), g["ld_exact"], rtol=1e-5)
    evals, evecs = kp.diagonalization()  # symeig by default for Kronecker products (:147-152)
    dense = np.stack([np.kron(K1[b].astype(np.float64), K2[b].astype(np.float64)) for b in range(2)])
    assert np.allclose(np.sort(evals.numpy(), -1), np.linalg.eigvalsh(dense), rtol=1e-4, atol=1e-5)
    q = evecs.to_dense().numpy()
    assert np.abs((q * evals.numpy()[..., None, :]) @ np.swapaxes(q, -1, -2) - dense).max() < 1e-3
    with settings.max_cholesky_size(0), pytest.raises(_hip.HipExtensionError):
        a1.solve(T(rhs))  # the solve needs the Kronecker matvec kernel: no silent ATen route
    with pytest.raises(_hip.HipExtensionError):
        lo.utils.minres(T(np.eye(8, dtype=np.float32)).matmul, torch.randn(8, 1))


def test_install_as_linear_operator_makes_the_package_a_drop_in():
    """`linear_operator_amd.install_as("linear_operator")`: GPyTorch-style imports resolve to this package, and the
    solver seam of the reference (rebinding `linear_operator.utils.linear_cg`, as
    linear_operator/test/linear_operator_test_case.py:555-556 does with mock.patch) is the seam `_solve` goes through.
    Runs in a subprocess: the alias must not leak into the other tests' sys.modules."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        from unittest import mock
        import torch
        import linear_operator_amd
        assert "linear_operator" not in sys.modules
        linear_operator_amd.install_as("linear_operator")
        import linear_operator
        from linear_operator import settings, operators, utils
        from linear_operator.operators import (AddedDiagLinearOperator, ConstantDiagLinearOperator, DiagLinearOperator,
                                               LowRankRootLinearOperator, TriangularLinearOperator)
        from linear_operator.operators.added_diag_linear_operator import AddedDiagLinearOperator as A2
        from linear_operator.utils.lanczos import lanczos_tridiag
        from linear_operator.utils.warnings import NumericalWarning
        import linear_operator.functions._inv_quad_logdet
        assert linear_operator is linear_operator_amd and A2 is AddedDiagLinearOperator
        assert linear_operator.utils is linear_operator_amd.utils
        assert issubclass(ConstantDiagLinearOperator, DiagLinearOperator) and issubclass(DiagLinearOperator, TriangularLinearOperator)
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(torch.randn(3, 40, 4)), DiagLinearOperator(torch.rand(3, 40) + 1))
        rhs = torch.randn(3, 40, 2)
        assert torch.allclose(torch.matmul(A, rhs), A.to_dense() @ rhs, atol=1e-4)      # __torch_function__ dispatch
        fake = mock.MagicMock(return_value=torch.zeros_like(rhs))
        with mock.patch("linear_operator.utils.linear_cg", new=fake), settings.max_cholesky_size(0):
            out = A.solve(rhs)                                                             # CG seam reached by name
        assert fake.call_count == 1 and torch.equal(out, torch.zeros_like(rhs))
        assert fake.call_args.args[0].__self__.__class__ is AddedDiagLinearOperator        # bound _matmul of the operator
        with settings.cg_tolerance(0.25):
            assert linear_operator_amd.settings.cg_tolerance.value() == 0.25               # one settings object
        try:
            linear_operator_amd.install_as("json")
        except ImportError:
            pass
        else:
            raise AssertionError("shadowing an imported package must be refused")
        print("OK")
    """ % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stderr[-3000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_small_closed_form_operators_keep_the_exact_branch_on_cpu(dtype):
    """`LowRankRoot + Diag` and `Kronecker + ConstantDiag` below `max_cholesky_size` on CPU / in fp64 are the
    plumbing case of the reference (functions/_solve.py:17-18): dense Cholesky, not the HIP closed forms."""
    g = torch.Generator().manual_seed(11)
    k1 = torch.randn(3, 3, generator=g, dtype=dtype)
    k2 = torch.randn(4, 4, generator=g, dtype=dtype)
    k1, k2 = k1 @ k1.T + torch.eye(3, dtype=dtype), k2 @ k2.T + torch.eye(4, dtype=dtype)
    a = lo.operators.KroneckerProductLinearOperator(lo.operators.DenseLinearOperator(k1), lo.operators.DenseLinearOperator(k2)) \
        + lo.operators.ConstantDiagLinearOperator(torch.tensor([0.5], dtype=dtype), 12)
    assert type(a).__name__ == "KroneckerProductAddedDiagLinearOperator"
    rhs = torch.randn(12, 2, generator=g, dtype=dtype)
    tol = 1e-4 if dtype == torch.float32 else 1e-10
    assert (a.to_dense() @ a.solve(rhs) - rhs).abs().max() < tol
    c, d = torch.randn(20, 3, generator=g, dtype=dtype), torch.rand(20, generator=g, dtype=dtype) + 0.5
    b = lo.operators.LowRankRootLinearOperator(c) + lo.operators.DiagLinearOperator(d)
    assert type(b).__name__ == "LowRankRootAddedDiagLinearOperator"
    rhs = torch.randn(20, 2, generator=g, dtype=dtype)
    assert (b.to_dense() @ b.solve(rhs) - rhs).abs().max() < tol
    assert (b.to_dense() @ b.solve(rhs[:, 0]) - rhs[:, 0]).abs().max() < tol  # vector right-hand side


def test_diag_and_triangular_inv_quad_logdet_follow_the_reference():
    """vector right-hand sides, empty tensors for terms that were not asked for
    (reference diag_linear_operator.py:161-191, triangular_linear_operator.py:181-204), `upper` of `_cholesky_solve`
    (triangular_linear_operator.py:72-89)."""
    g = torch.Generator().manual_seed(12)
    d = torch.rand(2, 9, generator=g) + 0.5
    D = lo.operators.DiagLinearOperator(d)
    v, m = torch.randn(2, 9, generator=g), torch.randn(2, 9, 3, generator=g)
    iq, ld = lo.operators.DiagLinearOperator(d[0]).inv_quad_logdet(v[0], logdet=True)
    assert torch.allclose(iq, (v[0] ** 2 / d[0]).sum()) and torch.allclose(ld, d[0].log().sum())
    iq, ld = D.inv_quad_logdet(m, logdet=False, reduce_inv_quad=False)
    assert iq.shape == (2, 3) and ld.numel() == 0 and torch.allclose(iq, (m ** 2 / d.unsqueeze(-1)).sum(-2))
    iq, ld = D.inv_quad_logdet(None, logdet=True)
    assert iq.numel() == 0 and torch.allclose(ld, d.log().sum(-1))
    cd = lo.operators.ConstantDiagLinearOperator(torch.tensor([2.0]), 5)
    iq, ld = cd.inv_quad_logdet(torch.ones(5), logdet=True)
    assert torch.allclose(iq, torch.tensor(2.5)) and torch.allclose(ld, torch.tensor(5 * np.log(2.0), dtype=torch.float32))
    k = torch.randn(6, 6, generator=g, dtype=torch.float64)
    k = k @ k.T + torch.eye(6, dtype=torch.float64)
    low = torch.linalg.cholesky(k)
    r = torch.randn(6, 2, generator=g, dtype=torch.float64)
    T = lo.operators.TriangularLinearOperator(low)
    U = lo.operators.TriangularLinearOperator(low.mT.contiguous(), upper=True)
    want = torch.linalg.solve(k, r)
    assert torch.allclose(T._cholesky_solve(r), want) and torch.allclose(U._cholesky_solve(r, upper=True), want)
    assert torch.allclose(T._cholesky_solve(r[:, 0]), want[:, 0])
    iq, ld = T.inv_quad_logdet(r[:, 0], logdet=True)
    assert iq.dim() == 0 and torch.allclose(iq, (r[:, 0] * torch.linalg.solve_triangular(low, r[:, :1], upper=False)[:, 0]).sum())
    assert torch.allclose(ld, low.diagonal().log().sum())
    iq, ld = T.inv_quad_logdet(r, logdet=False)
    assert ld.numel() == 0 and iq.dim() == 0
    neg = lo.operators.TriangularLinearOperator(-low)
    assert torch.isnan(neg.inv_quad_logdet(None, logdet=True)[1]) == (6 % 2 == 1)
    neg5 = lo.operators.TriangularLinearOperator(-low[:5, :5])
    assert torch.isnan(neg5.inv_quad_logdet(None, logdet=True)[1])


# ---------------------------------------------------------------- engine selection of lo_cg_solve_f32
def _plan(kind, N, R=32, c=1, k=15, pre="root+q", nt=0, B=512, x0=False, closure=False, max_iter=1000, cus=256, n2=0,
          kron_root=False, global_rule=False, diag_mode=_hip.LO_DIAG_FULL, const_pre=False, rs=False):
    """lo_cg_plan_f32 on a descriptor with placeholder (non-null) pointers: the plan reads shapes and null-ness only."""
    lib = _hip.load()
    FAKE = 0x1000
    op = _hip.OpDesc()
    op.kind, op.diag_mode, op.B, op.N, op.R, op.n2 = kind, diag_mode, B, N, R, n2
    op.A0, op.A1, op.d = FAKE, (FAKE if kind == _hip.LO_OP_KRON_DIAG else None), FAKE
    pd = None
    if pre and not closure:
        pd = _hip.PrecondDesc()
        pd.k, pd.ldq, pd.constant_diag, pd.dinv = k, 4 * (1 << max(0, ((k + 3) // 4 - 1).bit_length())), int(const_pre), FAKE
        pd.Q = FAKE if "q" in pre else None
        if "root" in pre:
            rf = 8 if R <= 8 else (16 if R <= 16 else 32)
            pd.F, pd.EF, pd.E, pd.rf_ld = FAKE, FAKE, FAKE, rf
            if rs:
                pd.RS = FAKE
        if kron_root:
            pd.kron_a, pd.kron_b, pd.kron_F = FAKE, FAKE, FAKE
    prm = _hip.CgParams()
    prm.c, prm.n_tridiag, prm.max_iter, prm.max_tridiag_iter, prm.floor_max_iter = c, nt, max_iter, 20, 0
    prm.tolerance, prm.eps, prm.stop_updating_after = 1e-4, 1e-10, 1e-10
    if global_rule:
        prm.stop_reduce = _hip.STOP_REDUCE_CB(lambda user, vals: 0)
    out = _hip.CgPlan()
    rc = lib.lo_cg_plan_f32(ctypes.byref(op), ctypes.byref(pd) if pd is not None else None, int(closure), int(x0),
                            ctypes.byref(prm), cus, ctypes.byref(out))
    assert rc == 0, rc
    return dict(resident=out.resident, iters=out.resident_iterations, ls=out.lockstep_cols, ls_gw=out.lockstep_group,
                serial=_hip.ENGINE_NAMES[out.serial_engine], gw=out.serial_group, lean=out.lean, needs_q=out.needs_q,
                stream=_hip.STREAM_PRE_NAMES[out.streaming_precond], chunk=out.poll_chunk, first_stop=out.first_stop_iteration,
                rspace=out.rspace)


LOW, DENSE, KRON, CB = _hip.LO_OP_LOWRANK_DIAG, _hip.LO_OP_DENSE_DIAG, _hip.LO_OP_KRON_DIAG, _hip.LO_OP_CALLBACK
ENGINE_TABLE = [
    # (label, arguments of _plan, expected subset of the plan)
    ("headline: cfg3 operator, one column", dict(kind=LOW, N=8192), dict(resident=1, iters=11, ls=0, serial="root", gw=8, lean=1, needs_q=0)),
    ("headline with the R-space form: iterations on R + 1 coordinates inside the resident launch", dict(kind=LOW, N=8192, rs=True),
     dict(resident=1, iters=11, ls=0, serial="root", gw=8, lean=1, rspace=2)),
    ("cfg3 with the R-space form: all columns in three streaming launches, lockstep + root as the repeat",
     dict(kind=LOW, N=8192, c=17, nt=16, rs=True), dict(resident=1, iters=21, ls=16, serial="root", lean=1, rspace=1)),
    ("R-space form, root form only, 17 columns", dict(kind=LOW, N=8192, c=17, nt=16, pre="root", rs=True),
     dict(resident=1, ls=0, serial="root", lean=1, rspace=1)),
    ("R-space form, one column WITH its tridiagonal: the streaming form records alpha / beta", dict(kind=LOW, N=8192, c=1, nt=1, rs=True),
     dict(resident=1, rspace=1, lean=1)),
    ("R-space form under the batch-global stop rule: state needed, not taken", dict(kind=LOW, N=8192, c=17, nt=16, rs=True, global_rule=True),
     dict(resident=1, rspace=0, lean=0)),
    ("R-space form, more than 32 columns", dict(kind=LOW, N=8192, c=40, rs=True), dict(resident=1, rspace=0)),
    ("R-space form, N = 16384, 17 columns", dict(kind=LOW, N=16384, c=17, nt=16, rs=True), dict(resident=1, rspace=1, ls=0, serial="root")),
    ("headline without the R-space form", dict(kind=LOW, N=8192), dict(rspace=0)),
    ("cfg2", dict(kind=LOW, N=8192, B=64), dict(resident=1, serial="root", gw=8, lean=1)),
    ("cfg3: 16 probes + rhs", dict(kind=LOW, N=8192, c=17, nt=16), dict(resident=1, iters=21, ls=16, ls_gw=8, serial="root", lean=1)),
    ("16 probes only", dict(kind=LOW, N=8192, c=16, nt=16), dict(resident=1, ls=16, serial="none", lean=1)),
    ("20 columns: chunks of 16 + 4", dict(kind=LOW, N=8192, c=20), dict(ls=20, serial="none")),
    ("19 columns: 16 lockstep + 3 serial", dict(kind=LOW, N=8192, c=19), dict(ls=16, serial="root")),
    ("Q form only, one column", dict(kind=LOW, N=8192, pre="q"), dict(resident=1, serial="gen2", gw=8, lean=0)),
    ("Q form only, 17 columns", dict(kind=LOW, N=8192, c=17, nt=16, pre="q"), dict(resident=1, ls=16, serial="gen2", lean=0)),
    ("root form only", dict(kind=LOW, N=8192, pre="root"), dict(resident=1, serial="root", needs_q=0, lean=1)),
    ("root form only, 17 columns: no lockstep without Q", dict(kind=LOW, N=8192, c=17, nt=16, pre="root"),
     dict(resident=1, ls=0, serial="root", lean=1, needs_q=0)),
    ("root form only beyond the resident kernels", dict(kind=LOW, N=100000, pre="root"), dict(resident=0, needs_q=1)),
    ("small members take small groups", dict(kind=LOW, N=1024), dict(resident=1, serial="root", gw=1)),
    ("N = 3000", dict(kind=LOW, N=3000, c=5, nt=4), dict(resident=1, ls=5, ls_gw=4, serial="none")),
    ("N = 16384", dict(kind=LOW, N=16384), dict(resident=1, serial="root", gw=16)),
    ("N = 16384, 17 columns: lockstep stops at 8192", dict(kind=LOW, N=16384, c=17, nt=16), dict(resident=1, ls=0, serial="root", gw=16)),
    ("N = 65536: groups of 64", dict(kind=LOW, N=65536), dict(resident=1, serial="root", gw=64)),
    ("N = 40000, Q form only: streaming, fused apply", dict(kind=LOW, N=40000, pre="q"), dict(resident=0, stream="fused_q", chunk=4)),
    ("N = 100000: streaming two-pass", dict(kind=LOW, N=100000), dict(resident=0, stream="two_pass")),
    ("N < 256", dict(kind=LOW, N=200, pre=None), dict(resident=0, stream="fused_cols_nopre")),
    ("no preconditioner", dict(kind=LOW, N=1500, pre=None), dict(resident=1, serial="root", gw=2, lean=1, stream="none")),
    ("thousands of tiny members keep the multi-launch step", dict(kind=DENSE, N=300, B=1000, c=11, nt=10, pre="q", k=5), dict(stream="two_pass")),
    ("200 members of 1000 rows, 11 columns", dict(kind=DENSE, N=1000, B=200, c=11, nt=10, pre="q", k=7), dict(stream="fused_cols")),
    ("no preconditioner, 9 columns", dict(kind=LOW, N=1500, pre=None, c=9), dict(resident=1, ls=9, serial="none")),
    ("rank-20 root is padded to 32", dict(kind=LOW, N=5000, R=20, c=20, nt=16), dict(resident=1, ls=20)),
    ("rank-8 root", dict(kind=LOW, N=2048, R=8, c=6), dict(resident=1, ls=6, ls_gw=2)),
    ("initial guess", dict(kind=LOW, N=8192, x0=True), dict(resident=0, stream="fused_q")),
    ("preconditioner closure", dict(kind=LOW, N=8192, closure=True), dict(resident=0, stream="closure", chunk=1)),
    ("max_iter below the floor", dict(kind=LOW, N=8192, max_iter=5), dict(resident=0, first_stop=4)),
    ("more than 64 columns", dict(kind=LOW, N=8192, c=70), dict(resident=0)),
    ("too few compute units", dict(kind=LOW, N=8192, cus=32), dict(resident=0)),
    ("batch-global stop rule over ranks", dict(kind=LOW, N=8192, global_rule=True), dict(resident=1, lean=0, chunk=1)),
    ("preconditioner rank 40: beyond the resident kernels", dict(kind=LOW, N=8192, k=40, pre="q"), dict(resident=0, stream="two_pass")),
    ("cfg4: Kronecker root form", dict(kind=KRON, N=65536, R=256, n2=256, B=128, kron_root=True, pre="q", const_pre=True,
                                       diag_mode=_hip.LO_DIAG_CONST), dict(resident=0, stream="fused_kron", chunk=4)),
    ("cfg4 without the Kronecker root form", dict(kind=KRON, N=65536, R=256, n2=256, B=128, pre="q", const_pre=True,
                                                  diag_mode=_hip.LO_DIAG_CONST), dict(resident=0, stream="fused_q")),
    ("Kronecker, three columns", dict(kind=KRON, N=65536, R=256, n2=256, B=128, c=3, pre="q"), dict(stream="two_pass")),
    ("Kronecker 128 x 128, three columns: one launch behind the product", dict(kind=KRON, N=16384, R=128, n2=128, B=16, c=3, pre="q"),
     dict(stream="fused_cols")),
    ("Kronecker 48 x 48: below the single-column fused apply", dict(kind=KRON, N=2304, R=48, n2=48, B=2, pre="q"), dict(stream="fused_cols")),
    ("cfg5: dense, 17 columns", dict(kind=DENSE, N=16384, c=17, nt=16, B=8, pre="q"), dict(resident=0, stream="fused_cols", first_stop=20)),
    ("dense, 17 columns, N = 20000: two row blocks per workgroup", dict(kind=DENSE, N=20000, c=17, nt=16, B=2, pre="q"), dict(stream="fused_cols")),
    ("dense, 17 columns, N = 70000: beyond the groups of 64", dict(kind=DENSE, N=70000, c=17, nt=16, B=1, pre="q"), dict(stream="two_pass")),
    ("dense, 17 columns, 40 members of 16384 rows: more than four rounds of the groups", dict(kind=DENSE, N=16384, c=17, nt=16, B=40, pre="q"),
     dict(stream="two_pass")),
    ("dense, 40 columns", dict(kind=DENSE, N=4000, c=40, B=1, pre="q"), dict(stream="two_pass")),
    ("dense, 11 columns, N = 4000, too few compute units for groups of 16", dict(kind=DENSE, N=4000, c=11, nt=10, B=1, pre="q", cus=32),
     dict(stream="two_pass")),
    ("dense, 11 columns, preconditioner rank 40", dict(kind=DENSE, N=4000, c=11, nt=10, B=1, pre="q", k=40), dict(stream="two_pass")),
    ("dense, one column, N = 16384: fused apply", dict(kind=DENSE, N=16384, B=8, pre="q"), dict(stream="fused_q")),
    ("dense, unpreconditioned", dict(kind=DENSE, N=1000, B=1, pre=None), dict(resident=0, stream="fused_cols_nopre")),
    ("closure operator", dict(kind=CB, N=8192, pre=None), dict(resident=0, chunk=1)),
]


@pytest.mark.parametrize("label,args,want", ENGINE_TABLE, ids=[t[0] for t in ENGINE_TABLE])
def test_cg_engine_selection_table(label, args, want):
    """Which engine `lo_cg_solve_f32` takes for which (kind, N, R, c, k, n_tridiag, preconditioner form, device size):
    `lo_cg_plan_f32` is the pure function the solver executes (csrc/lo_cg.hip: cg_plan), so an eligibility edit that moves
    a shape to a slower path shows up here, on a machine without a GPU."""
    got = _plan(**args)
    bad = {k: (got[k], v) for k, v in want.items() if got[k] != v}
    assert not bad, f"{label}: (got, want) {bad}\nfull plan: {got}"


def test_eigform_policy_builds_the_diagonal_form_when_the_solves_served_have_paid_for_it():
    """kernels._eigform_due (pure host logic): the ski-rental rule over the measured costs."""
    from linear_operator_amd import kernels as K

    old = K.EIGFORM_AFTER_USES
    try:
        K.EIGFORM_AFTER_USES = None
        first = {B: next(n for n in range(1, 10 ** 4) if K._eigform_due(n, B)) for B in (40, 64, 512, 4096)}
        assert first == {40: 134, 64: 137, 512: 25, 4096: 11}, first
        K.EIGFORM_AFTER_USES = 0
        assert K._eigform_due(1, 8)
        K.EIGFORM_AFTER_USES = 3
        assert not K._eigform_due(3, 512) and K._eigform_due(4, 512)
        K.EIGFORM_AFTER_USES = -1
        assert not K._eigform_due(10 ** 6, 4096)
    finally:
        K.EIGFORM_AFTER_USES = old
