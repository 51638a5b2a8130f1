"""The one-pass operator-resident matvec of low-rank + diagonal operators (csrc/lo_lowrank_mv.hip: k_lr_mv) through the C ABI
(lo_matvec_f32) and the operator API: RootLinearOperator._matmul (root_linear_operator.py:68-72) under
AddedDiagLinearOperator._matmul (added_diag_linear_operator.py:72-76).

Against the numpy oracle in float64 on the same seeded inputs (bar 2e-6 per column, the bar of the two-pass kernels), against
the two-pass kernels, bit for bit against itself, with every workgroup forced down the lost-hand-off path, and at the
north_star shape (512 x 8192 x 32) through size-independent properties.  Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest
import torch

import cases
from conftest import max_rel_err_cols
from oracle import lo_oracle as orc

pytestmark = pytest.mark.gpu

from linear_operator_amd import _hip  # noqa: E402
from linear_operator_amd import kernels as K  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


def _prof(fn):
    torch.cuda.synchronize()
    _hip.prof_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        return out, _hip.prof_report()
    finally:
        _hip.prof_enable(False)


@pytest.fixture
def env():
    """Set / restore the switches the library reads at every launch."""
    saved = {}

    def set_(name, value):
        saved.setdefault(name, os.environ.get(name))
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = value

    yield set_
    for name, value in saved.items():
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = value


# every group size (1, 2, 4, 8, 16, 32 workgroups per member), ragged members, all three padded ranks, one to four columns
# and the three diagonal modes
SHAPES = [
    (5, 300, 32, 1, "full"), (5, 1024, 32, 2, "const"), (5, 1025, 32, 1, "full"), (4, 2049, 20, 1, "none"),
    (3, 5000, 32, 2, "full"), (6, 8192, 32, 1, "full"), (2, 20000, 32, 1, "const"), (2, 32768, 17, 2, "full"),
    (7, 4097, 16, 4, "full"), (7, 3000, 9, 3, "const"), (9, 2500, 8, 4, "full"), (9, 8192, 5, 2, "none"),
    (70, 2048, 32, 1, "full"), (6, 8192, 32, 4, "full"), (5, 3001, 27, 3, "const"), (3, 16384, 32, 4, "none"),
]


@pytest.mark.parametrize("B,N,R,c,diag", SHAPES)
def test_resident_matvec_against_the_oracle_and_the_two_pass_kernels(B, N, R, c, diag, env):
    C, d, v = cases.lowrank_diag(4000 + N + R + c, B, N, R, c)
    if diag == "const":
        sig = (np.arange(B, dtype=np.float32) + 1.0) / 4.0
        desc = K.lowrank_diag_descriptor(dev(C), dev(sig), const_diag=True)
        d64 = np.broadcast_to(sig[:, None], d.shape).astype(np.float64)
    elif diag == "none":
        desc = K.lowrank_diag_descriptor(dev(C), None)
        d64 = np.zeros_like(d, dtype=np.float64)
    else:
        desc = K.lowrank_diag_descriptor(dev(C), dev(d))
        d64 = d.astype(np.float64)
    ref = orc.matvec_lowrank_diag(C.astype(np.float64), d64, v.astype(np.float64))
    vd = dev(v)
    y, prof = _prof(lambda: K.matvec(desc, vd))
    assert "lr_mv" in prof and "skinny_nn_R32" not in prof, prof  # the resident kernel is what ran
    assert max_rel_err_cols(host(y), ref) < 2e-6
    assert torch.equal(y, K.matvec(desc, vd))                      # fixed summation order: bit for bit
    env("LO_NO_RESIDENT_MV", "1")
    y2, prof2 = _prof(lambda: K.matvec(desc, vd))
    assert "lr_mv" not in prof2
    env("LO_NO_RESIDENT_MV", None)
    assert max_rel_err_cols(host(y), host(y2).astype(np.float64)) < 2e-6
    # every workgroup starts with the hand-off "lost": t = C^T v of the whole member recomputed from HBM by each workgroup
    env("LO_MV_TEST_FALLBACK", "1")
    y3 = K.matvec(desc, vd)
    env("LO_MV_TEST_FALLBACK", None)
    assert max_rel_err_cols(host(y3), ref) < 2e-6
    # and the launch after a lost one is a clean resident launch again (the error word is per launch)
    assert torch.equal(y, K.matvec(desc, vd))


def test_shapes_the_resident_kernel_does_not_take_run_the_two_pass_kernels():
    for (B, N, R, c) in [(3, 1000, 32, 5), (3, 1000, 8, 5), (3, 200, 32, 1), (2, 1000, 40, 1)]:
        C, d, v = cases.lowrank_diag(4100 + c + R, B, N, R, c)
        desc = K.lowrank_diag_descriptor(dev(C), dev(d))
        y, prof = _prof(lambda: K.matvec(desc, dev(v)))
        assert "lr_mv" not in prof, (B, N, R, c, prof)
        ref = orc.matvec_lowrank_diag(C.astype(np.float64), d.astype(np.float64), v.astype(np.float64))
        assert max_rel_err_cols(host(y), ref) < 2e-6


def test_operator_api_matmul_runs_the_resident_kernel():
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator

    C, d, v = cases.lowrank_diag(4200, 2 * 3, 3000, 32, 1)
    Cd, dd, vd = dev(C).reshape(2, 3, 3000, 32), dev(d).reshape(2, 3, 3000), dev(v).reshape(2, 3, 3000, 1)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cd), DiagLinearOperator(dd))
    y, prof = _prof(lambda: torch.matmul(A, vd))
    assert "lr_mv" in prof, prof
    ref = orc.matvec_lowrank_diag(C.astype(np.float64), d.astype(np.float64), v.astype(np.float64))
    assert max_rel_err_cols(host(y).reshape(6, 3000, 1), ref) < 2e-6
    A0 = AddedDiagLinearOperator(LowRankRootLinearOperator(Cd[0, 0]), DiagLinearOperator(dd[0, 0]))
    yv, prof_v = _prof(lambda: A0 @ vd[0, 0, :, 0])                # an unbatched operator and a 1-D right-hand side
    assert "lr_mv" in prof_v and yv.shape == (3000,) and torch.equal(yv, y[0, 0, :, 0])
    yr, prof_r = _prof(lambda: LowRankRootLinearOperator(Cd) @ vd)  # the root alone: no diagonal term
    assert "lr_mv" in prof_r
    ref_r = orc.matvec_lowrank_diag(C.astype(np.float64), np.zeros_like(d, dtype=np.float64), v.astype(np.float64))
    assert max_rel_err_cols(host(yr).reshape(6, 3000, 1), ref_r) < 2e-6


def test_north_star_shape_through_properties_and_float64_on_a_sample():
    """512 x 8192 x 32, one column (BASELINE.json's headline operator): linearity, the diagonal-only limit, symmetry of the
    bilinear form, and float64 on 16 members."""
    B, N, R = 512, 8192, 32
    g = torch.Generator().manual_seed(1234)
    C = (torch.randn(B, N, R, generator=g) / R ** 0.5).cuda()
    d = (torch.rand(B, N, generator=g) + 0.5).cuda()
    u = torch.randn(B, N, 1, generator=g).cuda()
    v = torch.randn(B, N, 1, generator=g).cuda()
    desc = K.lowrank_diag_descriptor(C, d)
    (yu, yv), prof = _prof(lambda: (K.matvec(desc, u), K.matvec(desc, v)))
    assert "lr_mv" in prof
    # linearity: A (2 u - 3 v) = 2 A u - 3 A v
    lin = K.matvec(desc, 2.0 * u - 3.0 * v)
    want = 2.0 * yu - 3.0 * yv
    assert float(((lin - want).norm(dim=-2) / want.norm(dim=-2)).max()) < 2e-6
    # symmetry: u^T (A v) = v^T (A u)
    uAv = (u.double() * yv.double()).sum(dim=(-2, -1))
    vAu = (v.double() * yu.double()).sum(dim=(-2, -1))
    scale = (u.double().norm(dim=(-2, -1)) * yv.double().norm(dim=(-2, -1)))
    assert float(((uAv - vAu).abs() / scale).max()) < 2e-6
    # a zero root leaves d o v exactly
    desc0 = K.lowrank_diag_descriptor(torch.zeros_like(C), d)
    assert torch.equal(K.matvec(desc0, v), d.unsqueeze(-1) * v)
    # float64 on a sample of members
    idx = torch.arange(0, B, 32, device="cuda")
    C64, d64, v64 = C[idx].double(), d[idx].double(), v[idx].double()
    ref = C64 @ (C64.mT @ v64) + d64.unsqueeze(-1) * v64
    assert float(((yv[idx].double() - ref).norm(dim=-2) / ref.norm(dim=-2)).max()) < 2e-6


def test_many_launches_keep_their_tags_apart():
    """The hand-off granules live in a buffer the library never clears between launches (tags grow from launch to launch):
    a few thousand launches of alternating shapes / group sizes against the first result of each shape."""
    cases_ = []
    for (B, N, R, c) in [(40, 8192, 32, 1), (64, 2048, 16, 2), (9, 20000, 32, 1), (300, 1024, 8, 4)]:
        C, d, v = cases.lowrank_diag(4300 + N, B, N, R, c)
        desc = K.lowrank_diag_descriptor(dev(C), dev(d))
        vd = dev(v)
        cases_.append((desc, vd, K.matvec(desc, vd)))
    for i in range(3000):
        desc, vd, y0 = cases_[i % len(cases_)]
        y = K.matvec(desc, vd)
        if i % 97 == 0:
            assert torch.equal(y, y0), i
    torch.cuda.synchronize()
    for desc, vd, y0 in cases_:
        assert torch.equal(K.matvec(desc, vd), y0)
