"""DenseLinearOperator: wraps a tensor (reference: linear_operator/operators/dense_linear_operator.py:11-123).
`_matmul` runs liblo_amd's dense matvec kernel for HIP fp32 tensors of the solver's shape; small / CPU
tensors (the N <= max_cholesky_size plumbing) use ATen like the reference."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import kernels as K
from ..utils.cholesky import cholesky_solve
from ._linear_operator import LinearOperator, to_dense


def _native_ok(t: Tensor, rhs: Tensor) -> bool:
    return t.is_cuda and rhs.is_cuda and t.dtype == torch.float32 and rhs.dtype == torch.float32 and rhs.dim() >= 2


class DenseLinearOperator(LinearOperator):
    def _check_args(self, tsr):
        if not torch.is_tensor(tsr):
            return "DenseLinearOperator must take a torch.Tensor; got {}".format(tsr.__class__.__name__)
        if tsr.dim() < 2:
            return "DenseLinearOperator expects a matrix (or batches of matrices) - got a Tensor of size {}.".format(
                tsr.shape
            )

    def __init__(self, tsr):
        super().__init__(tsr)
        self.tensor = tsr

    def _kernel_descriptor(self, batch_shape=None):
        t = self.tensor
        if not (t.is_cuda and t.dtype == torch.float32 and t.shape[-1] == t.shape[-2]):
            return None
        if batch_shape is not None and tuple(batch_shape) != tuple(t.shape[:-2]):
            t = t.expand(*batch_shape, *t.shape[-2:])
        return K.dense_diag_descriptor(t, None)

    def _cholesky_solve(self, rhs, upper: bool = False):
        return cholesky_solve(rhs, self.to_dense(), upper=upper)

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):  # reference :69-71
        res = K.bilinear_dense(left_vecs, right_vecs, self.batch_shape)
        return (_sum_to(res, self.tensor.shape),)

    def _diagonal(self) -> Tensor:
        return self.tensor.diagonal(dim1=-1, dim2=-2)

    def _expand_batch(self, batch_shape):
        return self.__class__(self.tensor.expand(*batch_shape, *self.matrix_shape))

    def _get_indices(self, row_index, col_index, *batch_indices) -> Tensor:
        return self.tensor[(*batch_indices, row_index, col_index)]

    def _matmul(self, rhs: Tensor) -> Tensor:
        t = self.tensor
        if _native_ok(t, rhs) and t.shape[-1] == t.shape[-2] and t.shape[-1] >= 256:
            batch = torch.broadcast_shapes(t.shape[:-2], rhs.shape[:-2])
            desc = self._kernel_descriptor(batch)
            return K.matvec(desc, rhs.expand(*batch, *rhs.shape[-2:]))
        return torch.matmul(t, rhs)

    def _t_matmul(self, rhs):
        return torch.matmul(self.tensor.mT, rhs)

    def _size(self) -> torch.Size:
        return self.tensor.size()

    def _transpose_nonbatch(self):
        return DenseLinearOperator(self.tensor.mT)

    def to_dense(self) -> Tensor:
        return self.tensor

    def __add__(self, other):
        if isinstance(other, DenseLinearOperator):
            return DenseLinearOperator(self.tensor + other.tensor)
        if isinstance(other, torch.Tensor):
            return DenseLinearOperator(self.tensor + other)
        return super().__add__(other)


def to_linear_operator(obj):
    """Tensor -> DenseLinearOperator; LinearOperator -> itself (reference :107-119)."""
    if torch.is_tensor(obj):
        return DenseLinearOperator(obj)
    if isinstance(obj, LinearOperator):
        return obj
    raise TypeError("object of class {} cannot be made into a LinearOperator".format(obj.__class__.__name__))


__all__ = ["DenseLinearOperator", "to_linear_operator", "to_dense"]


def _sum_to(grad: Tensor, shape) -> Tensor:
    """Sum a derivative computed at the broadcast batch shape down to the shape of the tensor it belongs to."""
    return grad if tuple(grad.shape) == tuple(shape) else grad.sum_to_size(*shape)
