"""SumLinearOperator / PsdSumLinearOperator (reference: operators/sum_linear_operator.py:16-116,
operators/psd_sum_linear_operator.py:10-18)."""
from __future__ import annotations

import torch
from torch import Tensor

from ._linear_operator import LinearOperator
from .dense_linear_operator import to_linear_operator


def _common_shape(shapes):
    """torch.broadcast_shapes, with the usual case (all terms already have one shape) answered without it: the torch
    helper costs ~15 us per call and the solve path builds several sums per call (detach, representation trees)."""
    first = shapes[0]
    if all(s == first for s in shapes[1:]):
        return torch.Size(first)
    return torch.broadcast_shapes(*shapes)


class SumLinearOperator(LinearOperator):
    def __init__(self, *linear_ops, **kwargs):
        try:
            linear_ops = tuple(to_linear_operator(lt) for lt in linear_ops)
        except TypeError:
            raise TypeError("All arguments of a SumLinearOperator should be LinearOperators or Tensors")
        batch_shape = _common_shape([lt.batch_shape for lt in linear_ops])
        linear_ops = tuple(lt._expand_batch(batch_shape) if lt.batch_shape != batch_shape else lt for lt in linear_ops)
        super().__init__(*linear_ops, **kwargs)
        self.linear_ops = linear_ops

    def _kernel_descriptor(self, batch_shape=None):
        """Lowering of the sum (reference `_matmul` :47-51, `_diagonal` :31-32): one structured term + at most one
        diagonal lowers like AddedDiag; 2 .. LO_MAX_TERMS structured terms (nested sums flattened, left to right)
        + at most one diagonal lower to an LO_OP_SUM descriptor -- matvec, CG, Lanczos, MINRES and the pivoted
        Cholesky then run on the device without per-term Python calls."""
        from .. import kernels as K
        from .diag_linear_operator import DiagLinearOperator

        batch_shape = torch.Size(self.batch_shape if batch_shape is None else batch_shape)
        flat = []

        def walk(op):
            if isinstance(op, SumLinearOperator):  # (AddedDiag is a sum of its operator and its diagonal)
                for sub in op.linear_ops:
                    walk(sub)
            else:
                flat.append(op)

        for op in self.linear_ops:
            walk(op)
        diags = [op for op in flat if isinstance(op, DiagLinearOperator)]
        others = [op for op in flat if not isinstance(op, DiagLinearOperator)]
        if len(diags) > 1 or not others:
            return None
        if len(others) == 1:
            return _attach_diag(others[0], diags[0] if diags else None, batch_shape)
        if len(others) > K._hip.LO_MAX_TERMS:
            return None
        terms = []
        for op in others:
            desc = op._kernel_descriptor(batch_shape)
            if desc is None or desc.diag_mode != 0 or desc.kind == K._hip.LO_OP_SUM:
                return None
            terms.append(desc)
        if len({(t.B, t.N) for t in terms}) != 1:
            return None
        return _sum_with_diag(K.sum_descriptor(terms), diags[0] if diags else None, batch_shape)

    def _diagonal(self) -> Tensor:
        return sum(op._diagonal() for op in self.linear_ops)

    def _expand_batch(self, batch_shape):
        return self.__class__(*[op._expand_batch(batch_shape) for op in self.linear_ops])

    def _get_indices(self, row_index, col_index, *batch_indices) -> Tensor:
        return sum(op._get_indices(row_index, col_index, *batch_indices) for op in self.linear_ops)

    def _matmul(self, rhs: Tensor) -> Tensor:  # reference :47-51
        if rhs.dim() >= 2 and rhs.is_cuda and rhs.dtype == torch.float32:
            desc = self._kernel_descriptor(torch.broadcast_shapes(self.batch_shape, rhs.shape[:-2]))
            if desc is not None:
                from .. import kernels as K

                return K.matvec(desc, rhs.expand(*desc.batch_shape, *rhs.shape[-2:]))
        return sum(op._matmul(rhs) for op in self.linear_ops)

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):  # reference :59-62
        return tuple(var for op in self.linear_ops for var in op._bilinear_derivative(left_vecs, right_vecs))

    def _t_matmul(self, rhs):
        return sum(op._t_matmul(rhs) for op in self.linear_ops)

    def _size(self) -> torch.Size:
        return _common_shape([op.shape for op in self.linear_ops])

    def _transpose_nonbatch(self):
        return self.__class__(*[op.mT for op in self.linear_ops])

    def to_dense(self) -> Tensor:
        return sum(op.to_dense() for op in self.linear_ops).contiguous()

    def __add__(self, other):  # reference :88-116
        from .added_diag_linear_operator import AddedDiagLinearOperator
        from .diag_linear_operator import DiagLinearOperator

        if isinstance(other, DiagLinearOperator):
            return AddedDiagLinearOperator(self, other)
        if isinstance(other, SumLinearOperator):
            return SumLinearOperator(*(list(self.linear_ops) + list(other.linear_ops)))
        if isinstance(other, LinearOperator):
            return SumLinearOperator(*(list(self.linear_ops) + [other]))
        if isinstance(other, Tensor):
            shape = torch.broadcast_shapes(self.shape, other.shape)
            new_self = self if shape == self.shape else self._expand_batch(shape[:-2])
            return SumLinearOperator(*(list(new_self.linear_ops) + [to_linear_operator(other.expand(shape))]))
        raise AttributeError("other must be a LinearOperator")


class PsdSumLinearOperator(SumLinearOperator):
    """A sum of positive semi-definite terms: samples add (reference psd_sum_linear_operator.py:15-18)."""

    def zero_mean_mvn_samples(self, num_samples: int) -> Tensor:
        return sum(op.zero_mean_mvn_samples(num_samples) for op in self.linear_ops)


def _attach_diag(base_op, diag_op, batch_shape):
    """Descriptor of `base_op (+ diag_op)` expanded to batch_shape, or None."""
    desc = base_op._kernel_descriptor(batch_shape)
    if desc is None or desc.diag_mode != 0:
        return None
    return _sum_with_diag(desc, diag_op, batch_shape)


def _sum_with_diag(desc, diag_op, batch_shape):
    """Attach a (Constant)DiagLinearOperator to a diagonal-free descriptor, or None if its tensor cannot be used."""
    from .diag_linear_operator import ConstantDiagLinearOperator

    if diag_op is None:
        return desc
    from .. import kernels as K

    if isinstance(diag_op, ConstantDiagLinearOperator):
        vals = diag_op.diag_values
        if not (vals.is_cuda and vals.dtype == torch.float32):
            return None
        return K._with_diag(desc, vals.expand(*batch_shape, 1)[..., 0], True)
    d = diag_op._diag
    if not (d.is_cuda and d.dtype == torch.float32):
        return None
    return K._with_diag(desc, d.expand(*batch_shape, d.shape[-1]), False)


__all__ = ["SumLinearOperator", "PsdSumLinearOperator"]
