"""lo_root_apply_add_f32 (csrc/lo_bilinear.hip: k_root_apply_add): out += U T in one pass over out -- the N-sized product of
the pull-back through the pivoted Cholesky of a root (functions/_pivoted_cholesky.py::_dense_root_vjp; the reference's
PivotedCholesky.backward, functions/_pivoted_cholesky.py:107-147) accumulated onto an existing gradient.  Against the
float64 product on the same operands (bar: 1e-6 relative to the size of the terms), the shapes the kernel does not take, and
the pull-back with and without the in-place path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from linear_operator_amd import _hip  # noqa: E402
from linear_operator_amd import kernels as K  # noqa: E402


@pytest.mark.parametrize("batch,N,D,R", [((3,), 1000, 15, 32), ((2, 2), 1024, 16, 32), ((5,), 257, 1, 8), ((1,), 4097, 32, 16),
                                         ((7,), 300, 9, 32), ((2,), 8192, 15, 32), ((4,), 1025, 20, 8), ((3,), 77, 5, 16)])
def test_root_apply_add_against_float64(batch, N, D, R):
    g = torch.Generator(device="cuda").manual_seed(N + D + R)
    U = torch.randn(*batch, N, D, generator=g, device="cuda")
    T = torch.randn(*batch, D, R, generator=g, device="cuda")
    out0 = torch.randn(*batch, N, R, generator=g, device="cuda")
    out = out0.clone()
    _hip.prof_enable(True)
    took = K.root_apply_add(U, T, out)
    torch.cuda.synchronize()
    prof = _hip.prof_report()
    _hip.prof_enable(False)
    assert took and "root_apply_add" in prof
    ref = out0.double() + U.double() @ T.double()
    scale = out0.double().abs() + U.double().abs() @ T.double().abs()
    assert float(((out.double() - ref).abs() / scale).max()) < 1e-6
    out2 = out0.clone()
    assert K.root_apply_add(U, T, out2) and torch.equal(out, out2)  # fixed order: bit for bit


def test_operands_the_kernel_does_not_take_are_left_alone():
    U = torch.randn(2, 100, 15, device="cuda")
    base = torch.randn(2, 100, 32, device="cuda")
    for T, out in ((torch.randn(2, 15, 20, device="cuda"), torch.randn(2, 100, 20, device="cuda")),   # R = 20
                   (torch.randn(2, 15, 32, device="cuda"), base[:, :, :].transpose(-1, -2).contiguous().transpose(-1, -2)),  # strided
                   (torch.randn(2, 15, 32, device="cuda").double(), base.double())):                 # float64
        keep = out.clone()
        assert not K.root_apply_add(U.to(T.dtype), T, out)
        assert torch.equal(out, keep)
    U40 = torch.randn(2, 100, 40, device="cuda")
    out = base.clone()
    assert not K.root_apply_add(U40, torch.randn(2, 40, 32, device="cuda"), out) and torch.equal(out, base)
    view = base.clone()[:1]  # a slice of a larger tensor's storage is not the caller's own
    assert not K.root_apply_add(U[:1], torch.randn(1, 15, 32, device="cuda"), view)


def test_pull_back_through_the_pivoted_cholesky_in_place_equals_the_sum():
    """pivoted_cholesky_vjp(..., accumulate_into=[g]) returns g itself, updated to g + (the pull-back), and agrees with
    the out-of-place result plus g."""
    from linear_operator_amd.functions._pivoted_cholesky import pivoted_cholesky_vjp
    from linear_operator_amd.operators import LowRankRootLinearOperator

    g = torch.Generator(device="cuda").manual_seed(5)
    B, N, R, m = 6, 1500, 32, 15
    C = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(C, None), m)
    GL = torch.randn(B, N, m, generator=g, device="cuda")
    op = LowRankRootLinearOperator(C)
    Lc = L.contiguous() if L.shape[-1] == m else L.mT.contiguous()
    (plain,) = pivoted_cholesky_vjp(op, perm, GL, factor=Lc)
    base = torch.randn(B, N, R, generator=g, device="cuda")
    acc = base.clone()
    (fused,) = pivoted_cholesky_vjp(op, perm, GL, factor=Lc, accumulate_into=[acc])
    assert fused is acc
    want = base.double() + plain.double()
    scale = want.abs().max()
    assert float((fused.double() - want).abs().max() / scale) < 1e-5
