"""Matmul Function: `Y = A R` with both pull-backs (reference behaviour: linear_operator/functions/_matmul.py:9-66).

    dR = A^T G   (reduced to R's own shape when R was broadcast over the operator's batch)
    dA = <G, dA R>  -> `_bilinear_derivative(G, R)` of the operator's representation (csrc/lo_bilinear.hip)
Right-hand sides are handled in column form throughout ([..., N, c]); a vector is a single column that is peeled
off again on the way out -- of the product and of its gradient.
"""
from __future__ import annotations

from torch.autograd import Function


def _as_columns(t):
    return t.unsqueeze(-1) if t.dim() == 1 else t


class Matmul(Function):
    @staticmethod
    def forward(ctx, representation_tree, rhs, *matrix_args):
        ctx.representation_tree, ctx.vector_rhs = representation_tree, rhs.dim() == 1
        cols = _as_columns(rhs)
        ctx.save_for_backward(cols, *matrix_args)
        product = representation_tree(*matrix_args)._matmul(cols)
        return product[..., 0] if ctx.vector_rhs else product

    @staticmethod
    def backward(ctx, grad_output):
        cols, *matrix_args = ctx.saved_tensors
        want_rhs, want_args = ctx.needs_input_grad[1], any(ctx.needs_input_grad[2:])
        d_rhs, d_args = None, (None,) * len(matrix_args)
        if want_rhs or want_args:
            g = _as_columns(grad_output)
            linear_op = ctx.representation_tree(*matrix_args)
            if want_args:
                d_args = tuple(linear_op._bilinear_derivative(g, cols))
            if want_rhs:
                # a right-hand side with fewer (or size-1) batch dimensions than the operator was expanded in the
                # forward product: its gradient is the sum over the expanded dimensions
                d_rhs = linear_op._t_matmul(g.contiguous()).sum_to_size(cols.shape)
                if ctx.vector_rhs:
                    d_rhs = d_rhs[..., 0]
        return (None, d_rhs, *d_args)
