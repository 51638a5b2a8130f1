"""RootLinearOperator R R^T and LowRankRootLinearOperator (reference: operators/root_linear_operator.py:16-160,
operators/low_rank_root_linear_operator.py:11-64).  With a dense root the matvec  y = R (R^T v)  lowers to the
skinny two-pass kernels (csrc/lo_skinny.hip)."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import kernels as K
from ._linear_operator import LinearOperator
from .dense_linear_operator import DenseLinearOperator, to_linear_operator


class RootLinearOperator(LinearOperator):
    def __init__(self, root):
        root = to_linear_operator(root)
        super().__init__(root)
        self.root = root

    def _dense_root(self):
        return self.root.tensor if isinstance(self.root, DenseLinearOperator) else None

    def _kernel_descriptor(self, batch_shape=None):
        r = self._dense_root()
        if r is None or not (r.is_cuda and r.dtype == torch.float32):
            return None
        if batch_shape is not None and tuple(batch_shape) != tuple(r.shape[:-2]):
            r = r.expand(*batch_shape, *r.shape[-2:])
        return K.lowrank_diag_descriptor(r, None)

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):
        """Derivative w.r.t. the root tensor of sum_d u_d^T R R^T v_d = U (V^T R) + V (U^T R): what the reference's
        generic autograd version (_linear_operator.py:336-393) yields for `_matmul` = root._matmul(root._t_matmul(v))
        (:68-72)."""
        r = self._dense_root()
        if r is None:
            return super()._bilinear_derivative(left_vecs, right_vecs)
        if not r.requires_grad:
            return (None,)
        res = K.bilinear_root(r, left_vecs, right_vecs)
        return (res if tuple(res.shape) == tuple(r.shape) else res.sum_to_size(*r.shape),)

    def _diagonal(self) -> Tensor:  # reference :22-28
        r = self._dense_root()
        if r is not None:
            return (r ** 2).sum(-1)
        return super()._diagonal()

    def _expand_batch(self, batch_shape):
        if len(batch_shape) == 0:
            return self
        return self.__class__(self.root._expand_batch(batch_shape))

    def _get_indices(self, row_index, col_index, *batch_indices) -> Tensor:  # reference :37-50
        r = self.root.to_dense()
        left = r[(*batch_indices, row_index)]
        right = r[(*batch_indices, col_index)]
        return (left * right).sum(-1)

    def _matmul(self, rhs: Tensor) -> Tensor:  # reference :68-72
        desc = None
        if rhs.dim() >= 2 and rhs.is_cuda and rhs.dtype == torch.float32:
            desc = self._kernel_descriptor(torch.broadcast_shapes(self.batch_shape, rhs.shape[:-2]))
        if desc is not None:
            return K.matvec(desc, rhs.expand(*desc.batch_shape, *rhs.shape[-2:]))
        return self.root._matmul(self.root._t_matmul(rhs))

    def _t_matmul(self, rhs):
        return self._matmul(rhs)

    def root_decomposition(self, method=None):
        return self

    def _size(self) -> torch.Size:
        return torch.Size((*self.root.batch_shape, self.root.size(-2), self.root.size(-2)))

    def _transpose_nonbatch(self):
        return self

    def to_dense(self) -> Tensor:
        r = self.root.to_dense()
        return r @ r.mT

    def zero_mean_mvn_samples(self, num_samples: int) -> Tensor:
        """L eps with eps ~ N(0, I_k)  (reference _linear_operator.py:2778-2791 with covar_root = root)."""
        r = self.root.to_dense()
        base = torch.randn(*self.batch_shape, r.size(-1), num_samples, dtype=self.dtype, device=self.device)
        return r.matmul(base).permute(-1, *range(self.dim() - 1)).contiguous()


class LowRankRootLinearOperator(RootLinearOperator):
    """Marks the root as a genuine low-rank factor: `LowRankRoot + Diag` / `.add_diagonal` build a
    LowRankRootAddedDiagLinearOperator whose solve is Woodbury, not CG (reference
    low_rank_root_linear_operator.py:20-64).  The CG path is reached, as in the reference, by constructing
    AddedDiagLinearOperator(LowRankRootLinearOperator(C), DiagLinearOperator(d)) explicitly."""

    def add_diagonal(self, diag: Tensor):  # reference :20-50
        from .diag_linear_operator import ConstantDiagLinearOperator, DiagLinearOperator
        from .low_rank_root_added_diag_linear_operator import LowRankRootAddedDiagLinearOperator

        if not self.is_square:
            raise RuntimeError("add_diag only defined for square matrices")
        diag_shape = diag.shape
        if len(diag_shape) == 0:
            diag_tensor = ConstantDiagLinearOperator(diag.unsqueeze(-1), diag_shape=self.shape[-1])
        elif diag_shape[-1] == 1:
            diag_tensor = ConstantDiagLinearOperator(diag, diag_shape=self.shape[-1])
        else:
            try:
                expanded_diag = diag.expand(self.shape[:-1])
            except RuntimeError:
                raise RuntimeError(
                    "add_diag for LinearOperator of size {} received invalid diagonal of size {}.".format(
                        self.shape, diag_shape
                    )
                )
            diag_tensor = DiagLinearOperator(expanded_diag)
        return LowRankRootAddedDiagLinearOperator(self, diag_tensor)

    def __add__(self, other):  # reference :52-64
        from .diag_linear_operator import DiagLinearOperator
        from .low_rank_root_added_diag_linear_operator import LowRankRootAddedDiagLinearOperator

        if isinstance(other, DiagLinearOperator):
            return LowRankRootAddedDiagLinearOperator(self, other)
        return super().__add__(other)


__all__ = ["RootLinearOperator", "LowRankRootLinearOperator"]
