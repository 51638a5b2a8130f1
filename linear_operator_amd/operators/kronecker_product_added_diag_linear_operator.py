"""KroneckerProductAddedDiagLinearOperator: (K1 (x) ... (x) KP) + D (reference:
operators/kronecker_product_added_diag_linear_operator.py:51-316; SURVEY 8(f) rank 3 -- what the reference's default
`+` / `add_diagonal` routing builds for a Kronecker product, :100-145 of kronecker_product_linear_operator.py).

Constant diagonal sigma^2 I (the GP noise case): no CG at all.  Per-factor symmetric eigendecompositions K_i = Q_i L_i Q_i^T
(ATen plumbing in `settings._linalg_dtype_symeig`, n_i x n_i) give
    (K + sigma^2 I)^-1 r = Q (Q^T r / (lambda + sigma^2)),   Q = Q_1 (x) Q_2,  lambda = L_1 (x) L_2          (:153-161)
    logdet = sum log(lambda + sigma^2)                                                                        (:86-90)
and the two products with Q / Q^T over the N = prod n_i rows run on the Kronecker matvec kernels (csrc/lo_kron.hip).
Kronecker-structured diagonal D = D_1 (x) .. (x) D_P with the factor shapes of K (:94-128, :166-219; the multitask
noise model): K + D = D^1/2 (S + I) D^1/2 with S = (x)_i D_i^-1/2 K_i D_i^-1/2 = Q Lambda Q^T per factor, so
    (K + D)^-1 r = D^-1/2 Q ((Lambda + 1)^-1 (Q^T D^-1/2 r)),    logdet = sum log d + sum log(lambda + 1)
-- ONE formulation for the reference's two branches (factors with constant diagonals are the case where Q_i are the
eigenvectors of K_i themselves, :189-193).
Other diagonals follow the reference's last branch: the AddedDiag CG path, WITHOUT a preconditioner (:132-134).
`_root_decomposition` / `_root_inv_decomposition` (:224-294; round 6) return the lazy roots
`Q diag((lambda + sigma^2)^{+-1/2})` and `D^{+-1/2} Q diag((lambda + 1)^{+-1/2})` as `MatmulLinearOperator`s whose products
run on the Kronecker matvec kernels -- the N x N factor is never formed."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import _hip
from .added_diag_linear_operator import AddedDiagLinearOperator
from .diag_linear_operator import ConstantDiagLinearOperator, DiagLinearOperator
from .kronecker_product_linear_operator import KroneckerProductDiagLinearOperator, KroneckerProductLinearOperator
from .. import settings


class KroneckerProductAddedDiagLinearOperator(AddedDiagLinearOperator):
    def __init__(self, *linear_ops, preconditioner_override=None):
        super().__init__(*linear_ops, preconditioner_override=preconditioner_override)
        if len(linear_ops) > 2:
            raise RuntimeError("An AddedDiagLinearOperator can only have two components")
        elif isinstance(linear_ops[0], DiagLinearOperator):
            self.diag_tensor, self.linear_op = linear_ops[0], linear_ops[1]
        elif isinstance(linear_ops[1], DiagLinearOperator):
            self.diag_tensor, self.linear_op = linear_ops[1], linear_ops[0]
        else:
            raise RuntimeError(
                "One of the LinearOperators input to AddedDiagLinearOperator must be a DiagLinearOperator!"
            )
        self._diag_is_constant = isinstance(self.diag_tensor, ConstantDiagLinearOperator)
        self._eig_cache = None
        self._sym_cache = None

    # ------------------------------------------------------------------ eigendecomposition of the factors
    def _factor_eig(self):
        """(evals [*batch, N] in the symeig dtype, Q^T and Q as Kronecker operators in the operator's dtype); detached:
        used inside the autograd Functions' forward (their backward goes through `_bilinear_derivative`)."""
        if self._eig_cache is None:
            with torch.no_grad():
                evals, evecs = self._symeig_of_product(self.linear_op.detach())
            self._eig_cache = (evals, evecs._transpose_nonbatch(), evecs)
        return self._eig_cache

    @staticmethod
    def _symeig_of_product(op):
        """Eigenvalues in the symeig dtype when the operator is a Kronecker product (its factors are decomposed one
        by one); any other operator goes through its own `_symeig`."""
        if isinstance(op, KroneckerProductLinearOperator):
            return op._symeig(eigenvectors=True, symeig_dtype_evals=True)
        return op._symeig(eigenvectors=True)

    # ------------------------------------------------------------------ Kronecker-structured diagonal
    def _structured(self) -> bool:
        lt, dlt = self.linear_op, self.diag_tensor
        return (isinstance(dlt, KroneckerProductDiagLinearOperator) and isinstance(lt, KroneckerProductLinearOperator)
                and len(lt.linear_ops) == len(dlt.linear_ops)
                and all(a.shape[-1] == b.shape[-1] for a, b in zip(lt.linear_ops, dlt.linear_ops)))

    def _symmetrized_eig(self, detach: bool):
        """(D^-1/2 as a vector [*batch, N], eigenvalues of S [*batch, N], Q) in the symeig dtype; per-factor eigh."""
        dt = settings._linalg_dtype_symeig.value()
        roots, evals, evecs = [], None, []
        for k_op, d_op in zip(self.linear_op.linear_ops, self.diag_tensor.linear_ops):
            kd, dd = k_op.to_dense(), d_op._diag
            if detach:
                kd, dd = kd.detach(), dd.detach()
            ir = dd.to(dt).rsqrt()
            ev, q = torch.linalg.eigh(ir.unsqueeze(-1) * kd.to(dt) * ir.unsqueeze(-2))
            ev = ev.clamp_min(0.0)
            roots.append(DiagLinearOperator(ir))
            evals = ev if evals is None else (evals.unsqueeze(-1) * ev.unsqueeze(-2)).reshape(*ev.shape[:-1], -1)
            evecs.append(q)
        return KroneckerProductDiagLinearOperator(*roots)._diag, evals, evecs

    def _solve_structured(self, rhs: Tensor) -> Tensor:
        if self._sym_cache is None:
            with torch.no_grad():
                ir, evals, evecs = self._symmetrized_eig(detach=True)
                q = KroneckerProductLinearOperator(*[e.to(self.dtype) for e in evecs])
                self._sym_cache = (ir.to(self.dtype).unsqueeze(-1), (evals + 1.0).reciprocal().to(self.dtype).unsqueeze(-1),
                                   q._transpose_nonbatch(), q)
        ir, inv, q_t, q = self._sym_cache
        return ir * q._matmul(inv * q_t._matmul(ir * rhs))

    @property
    def _has_closed_form_solve(self) -> bool:  # functions/_solve._solve: eigendecomposition forms at every size
        return bool(self._diag_is_constant or self._structured())

    def _solve(self, rhs: Tensor, preconditioner=None, num_tridiag: int = 0):
        if not self._diag_is_constant and self._structured():
            return self._solve_structured(rhs)
        if not self._diag_is_constant:
            return super()._solve(rhs, preconditioner=preconditioner, num_tridiag=num_tridiag)
        if isinstance(self.linear_op, KroneckerProductLinearOperator) and len(self.linear_op.linear_ops) == 2:
            _hip.require_hip(rhs)  # Q and Q^T are applied by the Kronecker matvec kernel: no ATen route for this solve
        evals, q_t, q = self._factor_eig()
        sig = self.diag_tensor._diagonal().to(evals.dtype)  # [*batch, N]
        inv = (evals + sig).reciprocal().to(rhs.dtype).unsqueeze(-1)
        return q._matmul(inv * q_t._matmul(rhs))

    def _solve_preconditioner(self):
        return None

    def _preconditioner(self):
        # solves don't use CG (constant diagonal) so don't waste time computing it  (:132-134)
        return None, None, None

    def _logdet(self) -> Tensor:
        if self._diag_is_constant:  # :86-90, differentiable through the factors' eigh
            evals, _ = self._symeig_of_product(self.linear_op)
            return torch.log(evals + self.diag_tensor._diagonal().to(evals.dtype)).sum(dim=-1).to(self.dtype)
        if self._structured():  # :94-128, differentiable through the factors' eigh and the diagonal factors
            ir, evals, _ = self._symmetrized_eig(detach=False)
            return (torch.log1p(evals).sum(dim=-1) - 2.0 * ir.log().sum(dim=-1)).to(self.dtype)
        return super().inv_quad_logdet(logdet=True)[1]

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet: bool = False, reduce_inv_quad: bool = True):  # :68-82
        inv_quad_term = None
        if inv_quad_rhs is not None:
            inv_quad_term, _ = super().inv_quad_logdet(inv_quad_rhs=inv_quad_rhs, logdet=False,
                                                       reduce_inv_quad=reduce_inv_quad)
        logdet_term = self._logdet() if logdet else None
        return inv_quad_term, logdet_term

    # ------------------------------------------------------------------ lazy roots (:224-294)
    def _eig_root(self, power: float):
        """R with R R^T = (K + D)^(2 power) for the two closed forms above, as a lazy product; None: no closed form."""
        from .matmul_linear_operator import MatmulLinearOperator

        if self._diag_is_constant:  # :227-230 / :263-266: Q diag((lambda + sigma^2)^power), Q = Q_1 (x) ... (x) Q_P
            evals, _, q = self._factor_eig()
            sig = self.diag_tensor._diagonal().to(evals.dtype)
            return MatmulLinearOperator(q, DiagLinearOperator((evals + sig).pow(power).to(self.dtype)))
        if self._structured():  # :234-256 / :270-292: ONE formulation for both branches, D^(+-1/2) Q diag((lambda + 1)^power)
            with torch.no_grad():
                ir, evals, evecs = self._symmetrized_eig(detach=True)
            q = KroneckerProductLinearOperator(*[e.to(self.dtype) for e in evecs])
            scale = ir.reciprocal() if power > 0 else ir  # D^(1/2) for the root, D^(-1/2) for the inverse root
            return MatmulLinearOperator(DiagLinearOperator(scale.to(self.dtype)),
                                        MatmulLinearOperator(q, DiagLinearOperator((evals + 1.0).pow(power).to(self.dtype))))
        return None

    def _root_decomposition(self):  # :224-258
        root = self._eig_root(0.5)
        return root if root is not None else super()._root_decomposition()

    def _root_inv_decomposition(self, initial_vectors=None, test_vectors=None):  # :260-294
        root = self._eig_root(-0.5)
        return root if root is not None else super()._root_inv_decomposition(initial_vectors=initial_vectors,
                                                                             test_vectors=test_vectors)

    def _choose_root_method(self) -> str:
        # the closed forms hold at every size: the exact branch of small operators (dense Cholesky of N x N) is not needed
        if self._diag_is_constant or self._structured():
            return "lanczos"  # (the name of the branch that calls `_root_decomposition`, reference :2190-2199)
        return super()._choose_root_method()

    def _symeig(self, eigenvectors: bool = False, return_evals_as_lazy: bool = False):  # :296-308
        if self._diag_is_constant:
            evals, evecs = self.linear_op._symeig(eigenvectors=eigenvectors)
            return evals + self.diag_tensor.diag_values, evecs
        return super()._symeig(eigenvectors=eigenvectors)

    def __add__(self, other):  # :310-316
        if isinstance(other, ConstantDiagLinearOperator) and self._diag_is_constant:
            return KroneckerProductAddedDiagLinearOperator(self.linear_op, self.diag_tensor + other)
        return super().__add__(other)


__all__ = ["KroneckerProductAddedDiagLinearOperator"]
