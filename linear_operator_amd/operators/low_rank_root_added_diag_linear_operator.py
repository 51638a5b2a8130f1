"""LowRankRootAddedDiagLinearOperator: C C^T + D with the Woodbury closed forms instead of CG (reference:
operators/low_rank_root_added_diag_linear_operator.py:20-170; SURVEY 8(f) rank 3 -- what the reference's default `+`
routing builds for `LowRankRoot + Diag`).

On the HIP path the closed form is the SAME kernel sequence as the preconditioner cache of AddedDiagLinearOperator:
with the full root C in the place of the pivoted-Cholesky factor, `z = r/d - Q (Q^T r)` with
`Q = D^-1/2 W R^-1`, `R^T R = I + W^T W`, `W = D^-1/2 C` IS `(C C^T + D)^-1 r` (Woodbury), and the cached
`logdet P = 2 sum log|R_ii| + sum log d` is `logdet(C C^T + D)`; the capacitance matrix is factored in fp64
(csrc/lo_precond.hip), the reference factors it in the operator's dtype."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import kernels as K
from ..utils.cholesky import psd_safe_cholesky
from .added_diag_linear_operator import AddedDiagLinearOperator
from .diag_linear_operator import DiagLinearOperator
from .root_linear_operator import LowRankRootLinearOperator


class LowRankRootAddedDiagLinearOperator(AddedDiagLinearOperator):
    def __init__(self, *linear_ops, preconditioner_override=None):
        if len(linear_ops) > 2:
            raise RuntimeError("An AddedDiagLinearOperator can only have two components")
        a, b = linear_ops
        if isinstance(a, DiagLinearOperator) and not isinstance(b, LowRankRootLinearOperator):
            raise RuntimeError(
                "A LowRankRootAddedDiagLinearOperator can only be created with a LowRankLinearOperator base!"
            )
        elif isinstance(b, DiagLinearOperator) and not isinstance(a, LowRankRootLinearOperator):
            raise RuntimeError(
                "A LowRankRootAddedDiagLinearOperator can only be created with a LowRankLinearOperator base!"
            )
        super().__init__(*linear_ops, preconditioner_override=preconditioner_override)
        self._woodbury_cache = None

    # ------------------------------------------------------------------ closed form on the device
    def _root_and_diag(self):
        C = self._linear_op.root.to_dense()
        batch = torch.broadcast_shapes(C.shape[:-2], self._diag_tensor.batch_shape)
        C = C.expand(*batch, *C.shape[-2:])
        d = self._diag_tensor._diagonal().expand(*batch, C.shape[-2])
        return C, d

    def _woodbury_factor(self):
        """WoodburyPreconditioner built from the FULL root: its apply is the exact inverse, its logdet the exact logdet."""
        if self._woodbury_cache is None:
            C, d = self._root_and_diag()
            if not (C.is_cuda and C.dtype == torch.float32):
                raise K._hip.HipExtensionError("the Woodbury closed form runs in liblo_amd: fp32 HIP tensors only")
            if C.shape[-1] > 32:
                raise K._hip.HipExtensionError("LowRankRootAddedDiagLinearOperator: root rank > 32 is not supported")
            first = d[..., :1]
            const = bool(torch.equal(d, first.expand_as(d)))
            d_arg = first[..., 0].contiguous() if const else d.contiguous()
            self._woodbury_cache = K.precond_build(C.contiguous(), d_arg, const)
        return self._woodbury_cache

    @property
    def chol_cap_mat(self) -> Tensor:  # reference :36-47 (small R x R factor; kept for API compatibility)
        C, d = self._root_and_diag()
        cap = torch.eye(C.shape[-1], dtype=C.dtype, device=C.device) + C.mT @ (C / d.unsqueeze(-1))
        return psd_safe_cholesky(cap)

    def _mul_constant(self, other):  # reference :49-57
        if other > 0:
            return self.__class__(self._linear_op._mul_constant(other), self._diag_tensor._mul_constant(other))
        return AddedDiagLinearOperator(self._linear_op._mul_constant(other), self._diag_tensor._mul_constant(other))

    def _preconditioner(self):  # reference :59-60
        return None, None, None

    def _solve_preconditioner(self):  # reference :91-92
        return None

    def _solve(self, rhs: Tensor, preconditioner=None, num_tridiag: int = 0):  # reference :62-89
        pre = self._woodbury_factor()
        batch = torch.broadcast_shapes(self.batch_shape, rhs.shape[:-2])
        return K.precond_apply(pre, rhs.expand(*batch, *rhs.shape[-2:]).contiguous())

    def _logdet(self) -> Tensor:  # reference :97-103
        return self._woodbury_factor().logdet

    def solve(self, right_tensor: Tensor, left_tensor=None) -> Tensor:
        """reference LinearOperator.solve (:2324-2379) -> Solve -> `_solve`.  The forward-only Solve Function would
        rebuild the operator from its leaf tensors and lose the factorisation cached here, so the closed form is
        called directly (same shape checks, same result)."""
        if not self.is_square:
            raise RuntimeError(
                "solve only operates on (batches of) square (positive semi-definite) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        if self.dim() == 2 and right_tensor.dim() == 1:
            if self.shape[-1] != right_tensor.numel():
                raise RuntimeError(
                    "LinearOperator (size={}) cannot be multiplied with right-hand-side Tensor (size={}).".format(
                        self.shape, right_tensor.shape
                    )
                )
        is_vec = right_tensor.dim() == 1
        x = self._solve(right_tensor.unsqueeze(-1) if is_vec else right_tensor)
        if left_tensor is not None:
            return left_tensor @ x
        return x.squeeze(-1) if is_vec else x

    def __add__(self, other):  # reference :105-115
        if isinstance(other, DiagLinearOperator):
            return self.__class__(self._linear_op, self._diag_tensor + other)
        return AddedDiagLinearOperator(self._linear_op + other, self._diag_tensor)

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):  # reference :117-170
        if not self.is_square:
            raise RuntimeError(
                "inv_quad_logdet only operates on (batches of) square (positive semi-definite) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        if inv_quad_rhs is not None:
            if self.dim() == 2 and inv_quad_rhs.dim() == 1:
                if self.shape[-1] != inv_quad_rhs.numel():
                    raise RuntimeError(
                        "LinearOperator (size={}) cannot be multiplied with right-hand-side Tensor (size={}).".format(
                            self.shape, inv_quad_rhs.shape
                        )
                    )
            elif self.dim() != inv_quad_rhs.dim():
                raise RuntimeError(
                    "LinearOperator (size={}) and right-hand-side Tensor (size={}) should have the same number "
                    "of dimensions.".format(self.shape, inv_quad_rhs.shape)
                )
            elif self.shape[-1] != inv_quad_rhs.shape[-2]:
                raise RuntimeError(
                    "LinearOperator (size={}) cannot be multiplied with right-hand-side Tensor (size={}).".format(
                        self.shape, inv_quad_rhs.shape
                    )
                )
        inv_quad_term, logdet_term = None, None
        if inv_quad_rhs is not None:
            rhs = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
            self_inv_rhs = self._solve(rhs)
            inv_quad_term = (self_inv_rhs * rhs).sum(dim=-2)
            if reduce_inv_quad:
                inv_quad_term = inv_quad_term.sum(dim=-1)
        if logdet:
            logdet_term = self._logdet()
        return inv_quad_term, logdet_term


__all__ = ["LowRankRootAddedDiagLinearOperator"]
