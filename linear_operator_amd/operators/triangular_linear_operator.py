"""TriangularLinearOperator: the class-identity anchor of the diagonal operators and the exact-solve plumbing of the
`N <= max_cholesky_size` branch (reference: linear_operator/operators/triangular_linear_operator.py:19-191;
`DiagLinearOperator` derives from it, diag_linear_operator.py:16, and GPyTorch tests `isinstance(op,
TriangularLinearOperator)`).

Not on the iterative hot path: products and solves of a dense triangular factor are plain ATen calls
(`torch.linalg.solve_triangular`), as in the reference.  Diagonal operators override everything with elementwise forms
and never call this constructor with a dense tensor.
"""
from __future__ import annotations

import torch
from torch import Tensor

from ._linear_operator import LinearOperator


class _TriangularLinearOperatorBase:
    """Marker base of all triangular operators (reference :19-22)."""


class TriangularLinearOperator(LinearOperator, _TriangularLinearOperatorBase):
    def __init__(self, tensor, upper: bool = False):
        if isinstance(tensor, TriangularLinearOperator):
            tensor = tensor._tensor
        if isinstance(tensor, LinearOperator):
            tensor = tensor.to_dense()
        super().__init__(tensor, upper=upper)
        self.upper = bool(upper)
        self._tensor = tensor

    # ---- operator protocol
    def _matmul(self, rhs: Tensor) -> Tensor:
        return self._tensor.matmul(rhs)

    def _t_matmul(self, rhs: Tensor) -> Tensor:
        return self._tensor.mT.matmul(rhs)

    def _size(self) -> torch.Size:
        return self._tensor.shape

    def _transpose_nonbatch(self):
        return TriangularLinearOperator(self._tensor.mT, upper=not self.upper)

    def _diagonal(self) -> Tensor:
        return self._tensor.diagonal(dim1=-2, dim2=-1)

    def _expand_batch(self, batch_shape):
        return TriangularLinearOperator(self._tensor.expand(*batch_shape, *self._tensor.shape[-2:]), upper=self.upper)

    def _get_indices(self, row_index, col_index, *batch_indices) -> Tensor:
        return self._tensor[(*batch_indices, row_index, col_index)]

    def to_dense(self) -> Tensor:
        return self._tensor

    # ---- exact algebra of a triangular factor
    def solve(self, right_tensor: Tensor, left_tensor=None) -> Tensor:
        """T^-1 R by substitution (reference :160-191)."""
        is_vec = right_tensor.dim() == 1
        cols = right_tensor.unsqueeze(-1) if is_vec else right_tensor
        res = torch.linalg.solve_triangular(self._tensor, cols, upper=self.upper)
        if is_vec:
            res = res.squeeze(-1)
        return res if left_tensor is None else left_tensor @ res

    def _solve(self, rhs: Tensor, preconditioner=None, num_tridiag: int = 0) -> Tensor:
        return self.solve(rhs)

    def _cholesky_solve(self, rhs: Tensor, upper: bool = False) -> Tensor:
        """(T T^H)^-1 rhs for `upper=False`, (T^H T)^-1 rhs for `upper=True`: two substitutions with this factor and
        its transpose; the ARGUMENT selects the product, as in the reference (:72-89)."""
        is_vec = rhs.dim() == 1
        cols = rhs.unsqueeze(-1) if is_vec else rhs
        t, t_is_upper = self._tensor, self.upper
        if upper:  # (T^H T)^-1 = T^-1 T^-H
            w = torch.linalg.solve_triangular(t.mT, cols, upper=not t_is_upper)
            res = torch.linalg.solve_triangular(t, w, upper=t_is_upper)
        else:  # (T T^H)^-1 = T^-H T^-1
            w = torch.linalg.solve_triangular(t, cols, upper=t_is_upper)
            res = torch.linalg.solve_triangular(t.mT, w, upper=not t_is_upper)
        return res.squeeze(-1) if is_vec else res

    def inverse(self) -> "TriangularLinearOperator":
        eye = torch.eye(self._tensor.size(-1), dtype=self._tensor.dtype, device=self._tensor.device)
        return TriangularLinearOperator(self.solve(eye.expand_as(self._tensor)), upper=self.upper)

    def logdet(self) -> Tensor:
        return self._diagonal().abs().log().sum(-1)

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        """Substitution for the quadratic form, the diagonal for the log-determinant (NaN for a negative determinant);
        a term that was not asked for comes back as an EMPTY tensor, never `None` (reference :181-204)."""
        empty = torch.empty(0, dtype=self.dtype, device=self.device)
        inv_quad_term = empty
        if inv_quad_rhs is not None:
            cols = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
            inv_quad_term = (cols * self.solve(cols)).sum(-2)
            if reduce_inv_quad or inv_quad_rhs.dim() == 1:
                inv_quad_term = inv_quad_term.sum(-1)
        logdet_term = empty
        if logdet:
            diag = self._diagonal()
            logdet_term = diag.abs().log().sum(-1)
            negative = torch.sign(diag).prod(-1) < 0
            logdet_term = torch.where(negative, torch.full_like(logdet_term, float("nan")), logdet_term)
        return inv_quad_term, logdet_term


__all__ = ["TriangularLinearOperator"]
