"""MatmulLinearOperator: the lazy product `left @ right` (reference: operators/matmul_linear_operator.py:27-139).

Only what the hot path's consumers need of it: it is the type of the roots that
`KroneckerProductAddedDiagLinearOperator._root_decomposition` / `_root_inv_decomposition` return
(kronecker_product_added_diag_linear_operator.py:224-294: `Q (x) ... @ diag(sqrt(lambda + sigma^2))` -- an N x N factor that
must never be formed densely at N = 65536), so it multiplies, transposes, reports its size and diagonal, and densifies for
small sizes.  Products with a Kronecker factor run on the Kronecker matvec kernels (csrc/lo_kron.hip) through the factor's
own `_matmul`."""
from __future__ import annotations

import torch
from torch import Tensor

from ._linear_operator import LinearOperator
from .dense_linear_operator import DenseLinearOperator, to_linear_operator
from .diag_linear_operator import DiagLinearOperator


class MatmulLinearOperator(LinearOperator):
    def __init__(self, left_linear_op, right_linear_op):
        left, right = to_linear_operator(left_linear_op), to_linear_operator(right_linear_op)
        if left.size(-1) != right.size(-2):
            raise RuntimeError(f"Incompatible dimensions for matmul: {tuple(left.shape)} and {tuple(right.shape)}")
        batch = torch.broadcast_shapes(left.batch_shape, right.batch_shape)
        if left.batch_shape != batch:
            left = left._expand_batch(batch)
        if right.batch_shape != batch:
            right = right._expand_batch(batch)
        super().__init__(left, right)
        self.left_linear_op, self.right_linear_op = left, right

    def _size(self) -> torch.Size:  # reference :131-132
        return torch.Size((*self.left_linear_op.batch_shape, self.left_linear_op.size(-2), self.right_linear_op.size(-1)))

    def _matmul(self, rhs: Tensor) -> Tensor:  # reference :96-100
        return self.left_linear_op._matmul(self.right_linear_op._matmul(rhs))

    def _t_matmul(self, rhs: Tensor) -> Tensor:  # reference :102-106
        return self.right_linear_op._t_matmul(self.left_linear_op._t_matmul(rhs))

    def _transpose_nonbatch(self):  # reference :134-137
        return self.__class__(self.right_linear_op._transpose_nonbatch(), self.left_linear_op._transpose_nonbatch())

    def _expand_batch(self, batch_shape):  # reference :49-54
        return self.__class__(self.left_linear_op._expand_batch(batch_shape), self.right_linear_op._expand_batch(batch_shape))

    def _diagonal(self) -> Tensor:  # reference :73-85
        lo, ro = self.left_linear_op, self.right_linear_op
        if isinstance(lo, DenseLinearOperator) and isinstance(ro, DenseLinearOperator):
            return (lo.tensor * ro.tensor.mT).sum(-1)
        if isinstance(lo, DiagLinearOperator) and isinstance(ro, DiagLinearOperator):
            return lo._diagonal() * ro._diagonal()
        return self.to_dense().diagonal(dim1=-1, dim2=-2)

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):  # reference :108-121
        if left_vecs.dim() == 1:
            left_vecs, right_vecs = left_vecs.unsqueeze(1), right_vecs.unsqueeze(1)
        left_grad = self.left_linear_op._bilinear_derivative(left_vecs, self.right_linear_op._matmul(right_vecs))
        right_grad = self.right_linear_op._bilinear_derivative(self.left_linear_op._t_matmul(left_vecs), right_vecs)
        return tuple(left_grad) + tuple(right_grad)

    def to_dense(self) -> Tensor:  # reference :139-141 (the right factor is densified, the left one multiplies it)
        return self.left_linear_op._matmul(self.right_linear_op.to_dense())


__all__ = ["MatmulLinearOperator"]
