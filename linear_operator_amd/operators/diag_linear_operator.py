"""DiagLinearOperator / ConstantDiagLinearOperator (reference: linear_operator/operators/diag_linear_operator.py:16-434).
Inside CG the diagonal is never applied on its own: AddedDiag lowers `A + D` to one fused kernel
(y = A v + d o v).  Stand-alone use is elementwise ATen, like the reference."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import kernels as K
from ._linear_operator import LinearOperator
from .triangular_linear_operator import TriangularLinearOperator


class DiagLinearOperator(TriangularLinearOperator):
    """(a diagonal matrix is triangular: class identity as in the reference, diag_linear_operator.py:16 -- every
    method of the triangular base is overridden below with its elementwise form)"""

    upper = False

    def __init__(self, diag: Tensor):
        LinearOperator.__init__(self, diag)  # (not the triangular constructor: there is no dense factor)
        self._diag = diag

    def __add__(self, other):  # reference :27-35
        if isinstance(other, DiagLinearOperator):
            return self.add_diagonal(other._diag)
        from .added_diag_linear_operator import AddedDiagLinearOperator

        return AddedDiagLinearOperator(other, self)

    def add_diagonal(self, added_diag: Tensor) -> "DiagLinearOperator":
        shape = torch.broadcast_shapes(self._diag.shape, added_diag.shape)
        return DiagLinearOperator(self._diag.expand(shape) + added_diag.expand(shape))

    def _diagonal(self) -> Tensor:
        return self._diag

    def _expand_batch(self, batch_shape):
        return self.__class__(self._diag.expand(*batch_shape, self._diag.size(-1)))

    def _get_indices(self, row_index, col_index, *batch_indices) -> Tensor:
        res = self._diag[(*batch_indices, row_index)]
        return res * torch.eq(row_index, col_index).to(device=res.device, dtype=res.dtype)

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):  # reference :37-45
        if not self._diag.requires_grad:
            return (None,)
        res = K.bilinear_diag(left_vecs, right_vecs, self.batch_shape)
        return (res if tuple(res.shape) == tuple(self._diag.shape) else res.sum_to_size(*self._diag.shape),)

    def _matmul(self, rhs: Tensor) -> Tensor:  # reference :203-230
        if rhs.ndimension() == 1:
            return self._diag * rhs
        return self._diag.unsqueeze(-1) * rhs

    def _t_matmul(self, rhs):
        return self._matmul(rhs)

    def _size(self) -> torch.Size:
        return self._diag.shape + self._diag.shape[-1:]

    def _transpose_nonbatch(self):
        return self

    def to_dense(self) -> Tensor:
        if self._diag.dim() == 0:
            return self._diag
        return torch.diag_embed(self._diag)

    def inverse(self):
        return self.__class__(self._diag.reciprocal())

    def logdet(self):
        return self._diag.log().sum(-1)

    def abs(self):
        return DiagLinearOperator(self._diag.abs())

    def sqrt(self):
        return DiagLinearOperator(self._diag.sqrt())

    def solve(self, right_tensor: Tensor, left_tensor=None) -> Tensor:
        res = self.inverse()._matmul(right_tensor)
        return left_tensor @ res if left_tensor is not None else res

    def _cholesky_solve(self, rhs: Tensor, upper: bool = False) -> Tensor:  # (D D^T)^-1 rhs, reference :47-48
        return rhs / self._diag.unsqueeze(-1).pow(2)

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        """Elementwise: sum_n rhs^2 / d and sum_n log d; terms that were not asked for are EMPTY tensors (reference
        :161-191).  The trailing dimensions of the right-hand side beyond [*batch, N] are its columns; as in the reference
        (:174-184) the final `.sum(-1)` of reduce_inv_quad is applied whatever is left -- for a [*batch, N] right-hand
        side of a batched operator that is the last BATCH dimension (a quirk of the reference, kept: drop-in)."""
        empty = torch.empty(0, dtype=self.dtype, device=self.device)
        inv_quad_term = empty
        if inv_quad_rhs is not None:
            n_cols = inv_quad_rhs.dim() - self._diag.dim()  # len(rhs.shape[1 + batch_dim:])
            diag = self._diag.reshape(*self._diag.shape, *([1] * max(n_cols, 0)))
            inv_quad_term = (inv_quad_rhs / diag * inv_quad_rhs).sum(-(1 + max(n_cols, 0)))
            if reduce_inv_quad:
                inv_quad_term = inv_quad_term.sum(-1)
        logdet_term = self._diag.log().sum(-1) if logdet else empty
        return inv_quad_term, logdet_term

    def zero_mean_mvn_samples(self, num_samples: int) -> Tensor:  # reference :273-277
        base = torch.randn(num_samples, *self._diag.shape, dtype=self.dtype, device=self.device)
        return base * self._diag.sqrt()


class ConstantDiagLinearOperator(DiagLinearOperator):
    """sigma I with diag_values [*batch, 1] (reference :303-434)."""

    def __init__(self, diag_values: Tensor, diag_shape: int):
        if diag_values.dim() == 0 or diag_values.size(-1) != 1:
            raise ValueError(
                f"diag_values argument to ConstantDiagLinearOperator needs to have a final singleton dimension. "
                f"Instead, got a value with shape {diag_values.shape}."
            )
        LinearOperator.__init__(self, diag_values, diag_shape=diag_shape)
        self.diag_values = diag_values
        self.diag_shape = diag_shape

    @property
    def _diag(self) -> Tensor:  # reference :346-350
        return self.diag_values.expand(*self.diag_values.shape[:-1], self.diag_shape)

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):  # reference :337-344
        if not self.diag_values.requires_grad:
            return (None,)
        res = K.bilinear_diag(left_vecs, right_vecs, self.batch_shape, constant=True)
        shape = self.diag_values.shape
        return (res if tuple(res.shape) == tuple(shape) else res.sum_to_size(*shape),)

    def __add__(self, other):
        if isinstance(other, ConstantDiagLinearOperator):
            if other.shape[-1] == self.shape[-1]:
                return ConstantDiagLinearOperator(self.diag_values + other.diag_values, self.diag_shape)
            raise RuntimeError(f"Trying to add constant diagonals of different sizes {self.shape} / {other.shape}")
        return super().__add__(other)

    def add_diagonal(self, added_diag: Tensor):
        if added_diag.dim() == 0 or added_diag.size(-1) == 1:
            d = added_diag if added_diag.dim() else added_diag.unsqueeze(-1)
            return ConstantDiagLinearOperator(self.diag_values + d, self.diag_shape)
        return DiagLinearOperator(self._diag + added_diag)

    def _expand_batch(self, batch_shape):
        return self.__class__(self.diag_values.expand(*batch_shape, 1), diag_shape=self.diag_shape)

    def _size(self) -> torch.Size:
        return torch.Size((*self.diag_values.shape[:-1], self.diag_shape, self.diag_shape))

    def inverse(self):
        return self.__class__(self.diag_values.reciprocal(), diag_shape=self.diag_shape)

    def abs(self):
        return self.__class__(self.diag_values.abs(), diag_shape=self.diag_shape)

    def sqrt(self):
        return self.__class__(self.diag_values.sqrt(), diag_shape=self.diag_shape)


__all__ = ["DiagLinearOperator", "ConstantDiagLinearOperator"]
