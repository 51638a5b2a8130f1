"""InvQuad Function: the column-wise quadratic forms diag(R^T A^-1 R) (reference behaviour:
linear_operator/functions/_inv_quad.py:10-93).  With S = A^-1 R saved from the forward and g the incoming gradient per
column:  d/dR = 2 S g,   d/dA = -(S g) S^T  -> `_bilinear_derivative` (csrc/lo_bilinear.hip)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from ._solve import _solve


class InvQuad(Function):
    @staticmethod
    def forward(ctx, representation_tree, inv_quad_rhs, *matrix_args):
        ctx.representation_tree = representation_tree
        ctx.is_vector = inv_quad_rhs.dim() == 1
        cols = inv_quad_rhs.unsqueeze(-1) if ctx.is_vector else inv_quad_rhs
        solves = _solve(representation_tree(*matrix_args), cols)
        ctx.save_for_backward(solves, *matrix_args)
        return (solves * cols).sum(dim=-2)

    @staticmethod
    def backward(ctx, grad_output):
        solves, *matrix_args = ctx.saved_tensors
        weighted = solves * grad_output.unsqueeze(-2)  # S g, one weight per column
        arg_grads = (None,) * len(matrix_args)
        if any(ctx.needs_input_grad[2:]):
            op = ctx.representation_tree(*matrix_args)
            arg_grads = tuple(op._bilinear_derivative(-weighted, solves))
        rhs_grad = 2.0 * weighted if ctx.needs_input_grad[1] else torch.zeros_like(solves)
        if ctx.is_vector:
            rhs_grad = rhs_grad.squeeze(-1)
        return (None, rhs_grad) + arg_grads
