"""k_pc_onchip_rows (csrc/lo_pivchol_onchip.hip): the resident pivoted Cholesky of dense and Kronecker operators
(reference functions/_pivoted_cholesky.py:56-101) -- every step of a member inside one launch, the running diagonal and
the permutation position of a row in registers, the pivot found through the tagged-granule exchange of the group.

  * pivots, permutation, number of steps and factor are BIT-identical to the streaming engine (LO_PC_NO_RESIDENT_ROWS=1:
    one `k_pc_update` + `k_pc_ctrl` per step), over group sizes 1 .. 64, ranks 1 .. 32, ragged N, more members than
    resident groups, Kronecker row sources;
  * the batch-global stopping rule (`torch.max(errors) > error_tol`, :57): members whose spectrum ends early keep
    stepping with the batch, the step count is the batch's;
  * against the numpy oracle: pivots exact, factor to rounding;
  * an injected hand-off timeout falls back to the streaming engine and returns its bits.
"""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from oracle import lo_oracle as orc  # noqa: E402  (the checker)


def _both(monkeypatch, desc, rank, tol=1e-3):
    monkeypatch.delenv("LO_PC_NO_RESIDENT_ROWS", raising=False)
    K._hip.prof_enable(True)
    L, perm = K.pivoted_cholesky(desc, rank, error_tol=tol)
    took = K._hip.prof_report()
    K._hip.prof_enable(False)
    assert any(k.startswith("pc_onchip_rows") for k in took), took  # the resident kernel is the one that ran
    monkeypatch.setenv("LO_PC_NO_RESIDENT_ROWS", "1")
    Ls, perms = K.pivoted_cholesky(desc, rank, error_tol=tol)
    monkeypatch.delenv("LO_PC_NO_RESIDENT_ROWS")
    return L, perm, Ls, perms


def _dense(seed, B, N, inner, jitter=0.05):
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.randn(B, N, inner, generator=g, device="cuda") / inner ** 0.5
    Kd = X @ X.mT
    Kd = (Kd + Kd.mT) * 0.5 + jitter * torch.eye(N, device="cuda")
    return Kd.contiguous()


@pytest.mark.parametrize("B,N,rank", [
    (1, 64, 5),        # one workgroup
    (3, 100, 7),       # ragged, one workgroup
    (1, 1000, 15),     # group of 4
    (2, 4097, 32),     # rank above 16: the wide layout of the L rows
    (1, 10001, 15),    # group of 64 with a ragged tail
    (40, 700, 10),     # several rounds of the resident groups
    (300, 256, 4),     # more members than resident groups, one workgroup each
    (1, 33000, 8),     # 4 GB operator, 64 workgroups of 516 rows
    (5, 2048, 1),      # a single step
])
def test_dense_resident_factorisation_is_the_streaming_one_bit_for_bit(monkeypatch, B, N, rank):
    Kd = _dense(500 + N + rank, B, N, 48)
    L, perm, Ls, perms = _both(monkeypatch, K.dense_diag_descriptor(Kd, None), rank, tol=1e-6)
    assert L.shape == Ls.shape == (B, N, rank)
    assert torch.equal(perm, perms) and torch.equal(L, Ls)
    # and it is a pivoted Cholesky: the Schur complement's diagonal is what is left (reference :93)
    left = Kd.diagonal(dim1=-2, dim2=-1) - (L * L).sum(-1)
    assert left.min().item() > -1e-4 * Kd.diagonal(dim1=-2, dim2=-1).max().item()


@pytest.mark.parametrize("B,n1,n2,rank", [(4, 64, 64, 15), (3, 100, 72, 9), (2, 200, 200, 15), (70, 33, 40, 20)])
def test_kronecker_resident_factorisation_is_the_streaming_one_bit_for_bit(monkeypatch, B, n1, n2, rank):
    K1, K2 = _dense(7, B, n1, n1, 0.1), _dense(8, B, n2, n2, 0.1)
    L, perm, Ls, perms = _both(monkeypatch, K.kron_diag_descriptor(K1, K2, None), rank, tol=1e-6)
    assert L.shape[-1] == rank and torch.equal(perm, perms) and torch.equal(L, Ls)


def test_batch_global_stop_rule_with_members_of_different_rank(monkeypatch):
    """Members of numerical rank 3 / 6 / 12 in one batch, error_tol 1e-3, rank 15 requested: the reference steps the whole
    batch until the LARGEST member error is below the tolerance (:57) -- the step count is the rank-12 member's, and the
    rows the finished members produce in between are the streaming engine's, bit for bit."""
    N = 900
    g = torch.Generator(device="cuda").manual_seed(11)
    mats = []
    for r in (3, 6, 12):
        X = torch.randn(N, r, generator=g, device="cuda") * 0.1
        mats.append(X @ X.mT + 1e-7 * torch.eye(N, device="cuda"))
    Kd = torch.stack(mats).contiguous()
    L, perm, Ls, perms = _both(monkeypatch, K.dense_diag_descriptor(Kd, None), 15, tol=1e-3)
    assert L.shape == Ls.shape and 12 <= L.shape[-1] < 15, L.shape
    assert torch.equal(perm, perms)
    assert torch.equal(torch.nan_to_num(L), torch.nan_to_num(Ls)) and torch.equal(L.isnan(), Ls.isnan())
    Lo, pivo = orc.pivoted_cholesky(orc.DenseRowSource(Kd.cpu().numpy()), 15, error_tol=1e-3)
    assert Lo.shape[-1] == L.shape[-1]
    assert np.array_equal(perm.cpu().numpy()[:, :3], pivo[:, :3])  # the pivots every member still resolves


@pytest.mark.parametrize("N,rank", [(1500, 15), (777, 32)])
def test_resident_factorisation_against_the_oracle(N, rank):
    B = 2
    M = np.stack([cases.spd_test_matrix(900 + i, N, dtype=np.float32) for i in range(B)])
    M = (M / np.abs(M).max()).astype(np.float32)
    L, perm = K.pivoted_cholesky(K.dense_diag_descriptor(torch.from_numpy(M).cuda(), None), rank, error_tol=1e-7)
    Lo, pivo = orc.pivoted_cholesky(orc.DenseRowSource(M), rank, error_tol=1e-7)
    m = Lo.shape[-1]
    assert L.shape[-1] == m
    assert np.array_equal(perm.cpu().numpy()[..., :m], pivo[..., :m])
    assert np.abs(L.cpu().numpy() - Lo).max() < 2e-5 * np.abs(Lo).max()


def test_injected_timeout_falls_back_to_the_streaming_engine(monkeypatch):
    Kd = _dense(99, 2, 1200, 48)
    desc = K.dense_diag_descriptor(Kd, None)
    monkeypatch.setenv("LO_PC_NO_RESIDENT_ROWS", "1")
    Ls, perms = K.pivoted_cholesky(desc, 12, error_tol=1e-6)
    monkeypatch.delenv("LO_PC_NO_RESIDENT_ROWS")
    monkeypatch.setenv("LO_OC_TEST_FALLBACK", "1")
    L, perm = K.pivoted_cholesky(desc, 12, error_tol=1e-6)
    monkeypatch.delenv("LO_OC_TEST_FALLBACK")
    assert torch.equal(L, Ls) and torch.equal(perm, perms)
