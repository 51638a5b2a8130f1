"""BASELINE.json cfg4 and cfg5 at their real per-member sizes: cfg4 at 256 (x) 256 iteration-pinned against the reference
(golden g24; Q form and Kronecker root form), cfg5 at N = 16384 with injected probes against the oracle."""
import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, max_rel_err_cols, tridiag_block_err

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from linear_operator_amd import settings  # noqa: E402
from linear_operator_amd.operators import (  # noqa: E402
    AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator, DiagLinearOperator,
    KroneckerProductLinearOperator, LowRankRootLinearOperator,
)
from oracle import lo_oracle as orc  # noqa: E402  (the checker)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


class ProbedAddedDiag(AddedDiagLinearOperator):
    _probes = None

    def _probe_vectors_and_norms(self):  # hook: reference _linear_operator.py:629-633
        return self._probes


def _precond(desc, d_t, const=False):
    L, perm = K.pivoted_cholesky(desc, 15)
    if desc.kind == K._hip.LO_OP_LOWRANK_DIAG and desc.R <= 32:
        return K.precond_build(L, d_t, constant_diag=const, root=desc.A0, perm=perm)
    return K.precond_build(L, d_t, constant_diag=const)


@pytest.mark.parametrize("form", ["api", "q_form"])
def test_cfg4_real_factor_size_iteration_pinned(form, monkeypatch):
    """Golden g24: the reference's iterate after exactly its 137 iterations at 256 (x) 256 (N = 65536, B = 2); the HIP path
    runs the same count (tolerance 0, max_iter = 137) and agrees per column to 1e-4.  `api`: AddedDiag(Kron, ConstantDiag)
    .solve through the operator API (fused Kronecker matvec + Kronecker root form when the build accepts it);
    `q_form`: the streaming Q-form preconditioner with the two-launch matvec."""
    g = load_golden("g24_kron256_iteration_pinned")
    K1, K2, sig, rhs = cases.kron_factors(2401, 2, 256, 256, 1)
    its = int(g["iterations"])
    if form == "api":
        A = AddedDiagLinearOperator(KroneckerProductLinearOperator(DenseLinearOperator(dev(K1)), DenseLinearOperator(dev(K2))),
                                    ConstantDiagLinearOperator(dev(sig), 65536))
        assert type(A) is AddedDiagLinearOperator
        import warnings
        with settings.cg_tolerance(0.0), settings.max_cg_iterations(its), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x = A.solve(dev(rhs))
    else:
        monkeypatch.setenv("LO_NO_KRON_ROOT", "1")
        monkeypatch.setenv("LO_NO_KRON_FUSED", "1")
        d = dev(sig[:, 0])
        desc = K.kron_diag_descriptor(dev(K1), dev(K2), d, const_diag=True)
        L, perm = K.pivoted_cholesky(desc.without_diag(), 15, contiguous=False)
        pre = K.precond_build(L, d, True)
        res = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=0.0, max_iter=its)
        assert res.iterations == its
        x = res.x
    assert max_rel_err_cols(host(x), g["x_pinned"]) < 1e-4


def test_cfg5_real_size_injected_probes_against_the_oracle():
    """BASELINE cfg5's operator at N = 16384 (one member, 16 probes + 1 right-hand side): the whole inv_quad_logdet
    pipeline -- pivoted Cholesky of the dense operator, preconditioner, 21 CG iterations on 17 columns with the
    16-wide matrix-core matvec, tridiagonals, SLQ -- against the numpy oracle on identical inputs and probes."""
    N, P = 16384, 16
    gen = torch.Generator(device="cuda").manual_seed(16384)
    X = torch.randn(1, N, N, generator=gen, device="cuda") / 128
    Kd = X @ X.mT
    Kd = ((Kd + Kd.mT) * 0.5).contiguous()
    del X
    d = torch.rand(1, N, generator=gen, device="cuda") + 0.5
    rhs = torch.randn(1, N, 1, generator=gen, device="cuda")
    Z = torch.randn(1, N, P, generator=gen, device="cuda")
    Zn = Z.norm(dim=-2, keepdim=True)
    Z = Z / Zn
    A = ProbedAddedDiag(DenseLinearOperator(Kd), DiagLinearOperator(d))
    A._probes = (Z, Zn)
    with settings.cg_tolerance(1e-4):
        iq, ld = A.inv_quad_logdet(rhs, logdet=True)
    # kernel level on the same inputs: pivots, solves, tridiagonals
    desc = K.dense_diag_descriptor(Kd, d)
    L, perm = K.pivoted_cholesky(desc, 15)
    pre = K.precond_build(L, d, constant_diag=False)
    res = K.cg_solve(desc, torch.cat([Z, rhs], -1).contiguous(), precond=pre, n_tridiag=P, tolerance=1e-4)
    Kh, dh, rh, Zh = host(Kd), host(d), host(rhs), host(Z)
    iqo, ldo, so, to, info, po = orc.inv_quad_logdet(lambda v: orc.matvec_dense_diag(Kh, dh, v), orc.DenseRowSource(Kh),
                                                     dh, rh, Zh, tolerance=1e-4)
    _, pivo = orc.pivoted_cholesky(orc.DenseRowSource(Kh), 15)
    assert np.array_equal(host(perm)[..., :15], pivo[..., :15]), "pivots differ from the oracle"
    assert res.iterations == info.iterations == 21
    assert max_rel_err_cols(host(res.x), so) < 1e-4
    assert np.allclose(host(pre.logdet), po.logdet, rtol=1e-5)
    assert np.allclose(host(iq), iqo[..., 0], rtol=1e-4, atol=0)
    assert np.allclose(host(ld), ldo, rtol=1e-4, atol=0), (host(ld), ldo)
    _, _, pinvk = K.tridiag_eigh_slq(res.t_mat, N)
    assert np.allclose(host(pinvk) + host(pre.logdet), ldo, rtol=1e-4, atol=0)
    # tridiagonals entry by entry on the leading block where the oracle's own coupling is still meaningful
    t, t_o = host(res.t_mat).astype(np.float64), to.astype(np.float64)
    k = min(t.shape[-1], t_o.shape[-1])
    off = np.abs(np.diagonal(t_o[..., :k, :k], 1, -2, -1))
    lead = int(min(np.argmax(np.concatenate([off, np.zeros_like(off[..., :1])], -1) <= 1e-3 * np.abs(t_o).max(), axis=-1).min(), 12))
    assert lead >= 4
    blk = t_o[..., :lead, :lead]
    assert (np.abs(t[..., :lead, :lead] - blk) / (np.abs(blk) + 1e-2 * np.abs(blk).max())).max() < 1e-3
