"""Seeded random shape sweeps of the HIP path against the oracle / exact fp64 values: odd sizes, ranks and column counts
that the hand-picked parity cases do not visit (ragged tails of every tile size, ranks that are not multiples of 4,
rank-deficient operators, batch 1).  Every case prints its parameters in the assertion message."""
import random

import numpy as np
import pytest
import torch

import cases
from conftest import max_rel_err_cols
from oracle import lo_oracle as orc

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


def test_sweep_pivoted_cholesky_bit_exact_on_ragged_shapes():
    """Pivots, permutation and factor bit-identical to the oracle for random (B, N, R, rank) incl. N not a multiple of
    any tile, rank > numerical rank (early stop of the whole batch), and the three row sources."""
    rnd = random.Random(11)
    for case in range(14):
        B = rnd.choice([1, 2, 5])
        N = rnd.choice([97, 256, 1000, 1537, 2048, 4099])
        R = rnd.choice([1, 3, 8, 13, 32])
        rank = rnd.choice([1, 4, 15, 16])
        C = cases.lowrank_diag(5000 + case, B, N, R, 1)[0]
        L, piv = K.pivoted_cholesky(K.lowrank_diag_descriptor(dev(C), None), rank)
        Lo, pivo = orc.pivoted_cholesky(orc.LowRankRowSource(C), rank)
        tag = f"lowrank B={B} N={N} R={R} rank={rank}"
        assert np.array_equal(host(piv), pivo), tag
        assert host(L).shape == Lo.shape and np.array_equal(host(L), Lo), tag
    for case in range(4):
        B, N, rank = rnd.choice([1, 3]), rnd.choice([130, 515, 1025]), rnd.choice([3, 15])
        Kd = cases.dense_diag(5100 + case, B, N, 1)[0]
        L, piv = K.pivoted_cholesky(K.dense_diag_descriptor(dev(Kd), None), rank)
        Lo, pivo = orc.pivoted_cholesky(orc.DenseRowSource(Kd), rank)
        assert np.array_equal(host(piv), pivo) and np.array_equal(host(L), Lo), f"dense B={B} N={N} rank={rank}"
    for case in range(4):
        B, n1, n2, rank = rnd.choice([1, 2]), rnd.choice([5, 16, 33]), rnd.choice([7, 32]), rnd.choice([4, 15])
        K1, K2, _, _ = cases.kron_factors(5200 + case, B, n1, n2, 1)
        L, piv = K.pivoted_cholesky(K.kron_diag_descriptor(dev(K1), dev(K2), None), rank)
        Lo, pivo = orc.pivoted_cholesky(orc.KronRowSource(K1, K2), rank)
        assert np.array_equal(host(piv), pivo) and np.array_equal(host(L), Lo), f"kron B={B} {n1}x{n2} rank={rank}"


def test_sweep_cg_low_rank_operators_against_exact_solution():
    """Every engine the dispatcher can pick (serial resident, lockstep, streaming, fused preconditioner apply) at random
    shapes, against the fp64 Woodbury solution; the Lanczos tridiagonals must be finite."""
    rnd = random.Random(12)
    for case in range(36):
        B = rnd.choice([1, 2, 3, 7, 33])
        N = rnd.choice([1000, 1024, 1500, 3000, 4097, 8192, 9000, 12000, 16384, 20000, 33000])
        R = rnd.choice([1, 3, 5, 8, 12, 16, 20, 31, 32])
        c = rnd.choice([1, 2, 3, 5, 8, 16, 17, 20, 33])
        k = rnd.choice([0, 1, 4, 7, 15, 16])
        const = rnd.random() < 0.3
        ntri = rnd.choice([0, 0, min(c, 16)])
        g = torch.Generator(device="cuda")
        g.manual_seed(6000 + case)
        Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
        if const:
            d = (torch.rand(B, 1, generator=g, device="cuda") + 0.5).expand(B, N).contiguous()
        else:
            d = torch.rand(B, N, generator=g, device="cuda") + 0.5
        rhs = torch.randn(B, N, c, generator=g, device="cuda")
        pre = None
        if k > 0:
            L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), k, contiguous=False)
            darg = d[:, 0].contiguous() if const else d
            pre = K.precond_build(L, darg, const, root=Cm, perm=perm) if case % 2 else K.precond_build(L, darg, const)
        res = K.cg_solve(K.lowrank_diag_descriptor(Cm, d), rhs, precond=pre, n_tridiag=ntri, tolerance=1e-5,
                         max_iter=400)
        C64, d64, r64 = Cm.double(), d.double(), rhs.double()
        Cd = C64 / d64.unsqueeze(-1)
        cap = torch.eye(R, device="cuda", dtype=torch.float64) + C64.mT @ Cd
        exact = r64 / d64.unsqueeze(-1) - Cd @ torch.linalg.solve(cap, Cd.mT @ r64)
        err = ((res.x.double() - exact).norm(dim=-2) / exact.norm(dim=-2)).max().item()
        tag = f"B={B} N={N} R={R} c={c} k={k} const={const} ntri={ntri} iters={res.iterations} err={err:.2e}"
        assert err < 1e-4 and not res.nan_detected, tag
        assert ntri == 0 or bool(torch.isfinite(res.t_mat).all()), tag


def test_sweep_cg_dense_and_kronecker_operators_against_exact_solution():
    rnd = random.Random(13)
    for case in range(10):
        B, N, c = rnd.choice([1, 2, 5]), rnd.choice([130, 515, 1025, 2050]), rnd.choice([1, 3, 8, 17])
        k, ntri = rnd.choice([0, 7, 15]), rnd.choice([0, min(c, 16)])
        Kd, d, rhs = cases.dense_diag(7000 + case, B, N, c)
        desc = K.dense_diag_descriptor(dev(Kd), dev(d))
        pre = None
        if k:
            L, _ = K.pivoted_cholesky(K.dense_diag_descriptor(dev(Kd), None), k, contiguous=False)
            pre = K.precond_build(L, dev(d), False)
        res = K.cg_solve(desc, dev(rhs), precond=pre, n_tridiag=ntri, tolerance=1e-5, max_iter=600)
        A = Kd.astype(np.float64) + np.stack([np.diag(v) for v in d.astype(np.float64)])
        exact = np.linalg.solve(A, rhs.astype(np.float64))
        err = max_rel_err_cols(host(res.x), exact)
        assert err < 2e-4, f"dense B={B} N={N} c={c} k={k} ntri={ntri} iters={res.iterations} err={err:.2e}"
        y = host(K.matvec(desc, dev(rhs)))
        assert max_rel_err_cols(y, A @ rhs.astype(np.float64)) < 1e-5
    for case in range(10):
        B, n1, n2 = rnd.choice([1, 2, 4]), rnd.choice([5, 16, 33, 128]), rnd.choice([7, 32, 128])
        c, k = rnd.choice([1, 1, 3, 9]), rnd.choice([0, 15])
        K1, K2, sig, rhs = cases.kron_factors(7100 + case, B, n1, n2, c, sigma=0.05)
        desc = K.kron_diag_descriptor(dev(K1), dev(K2), dev(sig[:, 0]), const_diag=True)
        pre = None
        if k:
            L, _ = K.pivoted_cholesky(desc, k, contiguous=False)
            pre = K.precond_build(L, dev(sig[:, 0]), True)
        res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-5, max_iter=1500)
        A = np.stack([np.kron(K1[b].astype(np.float64), K2[b].astype(np.float64)) + sig[b, 0] * np.eye(n1 * n2)
                      for b in range(B)])
        exact = np.linalg.solve(A, rhs.astype(np.float64))
        err = max_rel_err_cols(host(res.x), exact)
        assert err < 5e-4, f"kron B={B} {n1}x{n2} c={c} k={k} iters={res.iterations} err={err:.2e}"
        y = host(K.matvec(desc, dev(rhs)))
        assert max_rel_err_cols(y, A @ rhs.astype(np.float64)) < 1e-5


def test_sweep_preconditioner_and_lanczos_against_oracle():
    """Woodbury preconditioner (apply + logdet) and the Lanczos tridiagonalisation at random shapes."""
    rnd = random.Random(14)
    for case in range(10):
        B, N, R = rnd.choice([1, 3]), rnd.choice([300, 1000, 2049]), rnd.choice([5, 16, 32])
        k, c, const = rnd.choice([1, 5, 15, 16]), rnd.choice([1, 4, 17]), rnd.random() < 0.4
        C, d, v = cases.lowrank_diag(8000 + case, B, N, R, c)
        if const:
            d = np.broadcast_to(d[:, :1], d.shape).copy()
        Lo, pivo = orc.pivoted_cholesky(orc.LowRankRowSource(C), k)
        pre_o = orc.Preconditioner(Lo.astype(np.float64), d.astype(np.float64))
        L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(dev(C), None), k, contiguous=False)
        pre = K.precond_build(L, dev(d[:, 0].copy()) if const else dev(d), const)
        z = host(K.precond_apply(pre, dev(v)))
        tag = f"B={B} N={N} R={R} k={k} c={c} const={const}"
        assert max_rel_err_cols(z, pre_o.apply(v.astype(np.float64))) < 5e-5, tag
        assert np.allclose(host(pre.logdet), pre_o.logdet, rtol=1e-5), tag
    for case in range(5):
        B, N, R, P, steps = rnd.choice([1, 2]), rnd.choice([257, 1000, 2048]), rnd.choice([8, 32]), rnd.choice([1, 4, 16]), rnd.choice([5, 12, 20])
        C, d, _ = cases.lowrank_diag(8100 + case, B, N, R, 1)
        V = cases.randn(8200 + case, B, N, P, dtype=np.float32)
        q, t = K.lanczos_tridiag(K.lowrank_diag_descriptor(dev(C), dev(d)), dev(V), steps)
        qo, to = orc.lanczos_tridiag(lambda x: orc.matvec_lowrank_diag(C, d, x), steps, V)
        tag = f"lanczos B={B} N={N} R={R} P={P} steps={steps}"
        assert tuple(t.shape) == to.shape and tuple(q.shape) == qo.shape, tag
        # early steps agree closely; later ones separate at the fp32 re-orthogonalisation noise of BOTH sides
        n0 = min(5, to.shape[-1])
        assert np.abs(host(t)[..., :n0, :n0] - to[..., :n0, :n0]).max() <= 2e-3 * np.abs(to).max(), tag
        # what the consumers rely on: Q orthonormal and Q T Q^T = A on the Krylov space (Q^T A Q = T)
        qh = host(q).astype(np.float64)
        gram = np.swapaxes(qh, -1, -2) @ qh
        assert np.abs(gram - np.eye(gram.shape[-1])).max() < 1e-4, tag


@pytest.mark.parametrize("N", [1024, 1100, 2048, 2049, 4096])
def test_small_members_take_small_groups(N):
    """Members of up to 1024 / 2048 / 4096 rows run the resident kernels in groups of 1 / 2 / 4 workgroups (a group of
    one has no hand-off at all): pivoted Cholesky bit-identical to the oracle, serial-column and 16-column lockstep CG
    with tridiagonals against the exact solution, and identical results with the groups of eight of `LO_OC_GW8`."""
    import os

    B, R = 37, 32
    C, d, _ = cases.lowrank_diag(9500 + N, B, N, R, 1)
    rhs = cases.randn(9600 + N, B, N, 17, dtype=np.float32)
    L, piv = K.pivoted_cholesky(K.lowrank_diag_descriptor(dev(C), None), 15)
    Lo, pivo = orc.pivoted_cholesky(orc.LowRankRowSource(C), 15)
    assert np.array_equal(host(piv), pivo) and np.array_equal(host(L), Lo)
    Lr, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(dev(C), None), 15, contiguous=False)
    pre = K.precond_build(Lr, dev(d), False, root=dev(C), perm=perm)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    C64, d64, r64 = C.astype(np.float64), d.astype(np.float64), rhs.astype(np.float64)
    Cd = C64 / d64[..., None]
    cap = np.eye(R) + np.swapaxes(C64, -1, -2) @ Cd
    exact = r64 / d64[..., None] - Cd @ np.linalg.solve(cap, np.swapaxes(Cd, -1, -2) @ r64)
    res17 = K.cg_solve(desc, dev(rhs), precond=pre, n_tridiag=16, tolerance=1e-5)       # lockstep + serial column
    res1 = K.cg_solve(desc, dev(rhs[..., :1].copy()), precond=pre, tolerance=1e-5)      # serial column alone
    assert max_rel_err_cols(host(res17.x), exact) < 1e-4 and bool(torch.isfinite(res17.t_mat).all())
    assert max_rel_err_cols(host(res1.x), exact[..., :1]) < 1e-4
    os.environ["LO_OC_GW8"] = "1"
    try:
        ref17 = K.cg_solve(desc, dev(rhs), precond=pre, n_tridiag=16, tolerance=1e-5)
        L8, piv8 = K.pivoted_cholesky(K.lowrank_diag_descriptor(dev(C), None), 15)
    finally:
        del os.environ["LO_OC_GW8"]
    assert res17.iterations == ref17.iterations
    assert max_rel_err_cols(host(res17.x), host(ref17.x).astype(np.float64)) < 1e-5
    assert torch.equal(L8, L) and torch.equal(piv8, piv)


def test_pivoted_cholesky_ranks_17_to_32_stay_resident_and_bit_exact():
    """`max_preconditioner_size` 17 .. 32: the L rows of a workgroup take 128 KB of LDS (one workgroup per CU) and the
    factorisation stays operator-resident; pivots, permutation and factor bit-identical to the oracle for every group
    size, ragged N, padded root ranks, and rank > numerical rank (whole-batch early stop)."""
    from linear_operator_amd import _hip

    rnd = random.Random(21)
    shapes = [(3, 256, 32, 17), (40, 1000, 20, 24), (5, 2048, 32, 32), (2, 4099, 8, 31), (33, 8192, 32, 20),
              (2, 12000, 16, 32), (1, 20000, 32, 17), (2, 1537, 3, 20), (300, 1024, 32, 32)]
    for case, (B, N, R, rank) in enumerate(shapes):
        C = cases.lowrank_diag(9800 + case, B, N, R, 1)[0]
        _hip.prof_enable(True)
        L, piv = K.pivoted_cholesky(K.lowrank_diag_descriptor(dev(C), None), rank)
        torch.cuda.synchronize()
        prof = _hip.prof_report()
        _hip.prof_enable(False)
        Lo, pivo = orc.pivoted_cholesky(orc.LowRankRowSource(C), rank)
        tag = f"B={B} N={N} R={R} rank={rank}"
        assert "pc_onchip" in prof, (tag, sorted(prof))
        assert np.array_equal(host(piv), pivo), tag
        assert host(L).shape == Lo.shape and np.array_equal(host(L), Lo), tag
    del rnd


def test_pivoted_cholesky_nan_column_of_an_exhausted_member():
    """A rank-deficient member (R = 11 < rank 32) whose remaining diagonal has gone negative by rounding: the reference
    takes sqrt(max) of a negative number and that member's column is NaN from then on, while the batch continues
    because other members are still above the tolerance.  The kernel (groups of 32, 128 KB of L rows per workgroup)
    reproduces it: same pivots, NaNs in the same places, every other entry bit-identical.  (Found by
    tools/fuzz_resident.py, seed 99.)"""
    B, N, R, rank = 130, 32768, 11, 32
    C = cases.lowrank_diag(693751327, B, N, R, 1)[0]
    L, piv = K.pivoted_cholesky(K.lowrank_diag_descriptor(dev(C), None), rank)
    with np.errstate(invalid="ignore"):
        Lo, pivo = orc.pivoted_cholesky(orc.LowRankRowSource(C), rank)
    assert np.isnan(Lo).any(), "the case no longer exercises the NaN column"
    assert np.array_equal(host(piv), pivo)
    assert np.array_equal(np.isnan(host(L)), np.isnan(Lo)) and np.array_equal(host(L), Lo, equal_nan=True)


@pytest.mark.usefixtures("legacy_resident_engines")
@pytest.mark.parametrize("N,c", [(40000, 1), (65536, 1), (50000, 3)])
def test_large_members_take_groups_of_64(N, c):
    """32768 < N <= 65536: the root-form resident CG runs a member on 64 workgroups (lane-parallel two-hop all-reduce)
    instead of falling to the streaming engine; against the exact fp64 Woodbury solution."""
    from linear_operator_amd import _hip

    B, R = 20, 32
    g = torch.Generator(device="cuda")
    g.manual_seed(9700 + N)
    Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    d = torch.rand(B, N, generator=g, device="cuda") + 0.5
    rhs = torch.randn(B, N, c, generator=g, device="cuda")
    L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
    pre = K.precond_build(L, d, False, root=Cm, perm=perm)
    desc = K.lowrank_diag_descriptor(Cm, d)
    _hip.prof_enable(True)
    res = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-5)
    torch.cuda.synchronize()
    prof = _hip.prof_report()
    _hip.prof_enable(False)
    assert "cg_onchip" in prof, sorted(prof)
    C64, d64, r64 = Cm.double(), d.double(), rhs.double()
    Cd = C64 / d64.unsqueeze(-1)
    cap = torch.eye(R, device="cuda", dtype=torch.float64) + C64.mT @ Cd
    exact = r64 / d64.unsqueeze(-1) - Cd @ torch.linalg.solve(cap, Cd.mT @ r64)
    err = ((res.x.double() - exact).norm(dim=-2) / exact.norm(dim=-2)).max().item()
    assert err < 1e-4, (N, c, res.iterations, err)


@pytest.mark.parametrize("n1,n2,c", [(64, 100, 1), (72, 68, 1), (64, 132, 3), (100, 64, 1), (76, 76, 9), (68, 96, 1)])
def test_kronecker_factors_of_any_multiple_of_four_use_the_matrix_core_engine(n1, n2, c):
    """Factor sizes that are multiples of 4 (>= 64) but not of the 128-wide tile: guarded tiles of the matrix-core GEMMs
    (lo_kron.hip, GUARD) -- matvec with / without diagonal and a preconditioned CG solve against the dense fp64 values."""
    from linear_operator_amd import _hip

    B = 3
    K1, K2, sig, rhs = cases.kron_factors(9800 + n1 + n2, B, n1, n2, c, sigma=0.05)
    desc = K.kron_diag_descriptor(dev(K1), dev(K2), dev(sig[:, 0]), const_diag=True)
    A = np.stack([np.kron(K1[b].astype(np.float64), K2[b].astype(np.float64)) for b in range(B)])
    _hip.prof_enable(True)
    y = host(K.matvec(desc, dev(rhs)))
    torch.cuda.synchronize()
    prof = _hip.prof_report()
    _hip.prof_enable(False)
    assert "kron_gemm_mfma" in prof, sorted(prof)
    Ad = A + np.stack([sig[b, 0] * np.eye(n1 * n2) for b in range(B)])
    assert max_rel_err_cols(y, Ad @ rhs.astype(np.float64)) < 1e-5
    y0 = host(K.matvec(K.kron_diag_descriptor(dev(K1), dev(K2), None), dev(rhs)))
    assert max_rel_err_cols(y0, A @ rhs.astype(np.float64)) < 1e-5
    L, _ = K.pivoted_cholesky(desc, 15, contiguous=False)
    pre = K.precond_build(L, dev(sig[:, 0]), True)
    res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-5, max_iter=1500)
    err = max_rel_err_cols(host(res.x), np.linalg.solve(Ad, rhs.astype(np.float64)))
    assert err < 5e-4, (n1, n2, c, res.iterations, err)
