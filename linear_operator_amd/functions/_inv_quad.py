"""InvQuad Function (reference: linear_operator/functions/_inv_quad.py:10-61): diag(R^T A^-1 R)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings


def _solve(linear_op, rhs):
    if settings.fast_computations.solves.off() or linear_op.size(-1) <= settings.max_cholesky_size.value():
        return linear_op.cholesky()._cholesky_solve(rhs)
    with torch.no_grad():
        preconditioner = linear_op.detach()._solve_preconditioner()
    return linear_op._solve(rhs, preconditioner)


class InvQuad(Function):
    @staticmethod
    def forward(ctx, representation_tree, *args):
        inv_quad_rhs, *matrix_args = args
        ctx.representation_tree = representation_tree
        linear_op = representation_tree(*matrix_args)
        ctx.is_vector = inv_quad_rhs.ndimension() == 1
        if ctx.is_vector:
            inv_quad_rhs = inv_quad_rhs.unsqueeze(-1)
        solves = _solve(linear_op, inv_quad_rhs)
        ctx.save_for_backward(*matrix_args, solves)
        return (solves * inv_quad_rhs).sum(-2)

    @staticmethod
    def backward(ctx, inv_quad_grad_output):  # reference :63-93
        *matrix_args, inv_quad_solves = ctx.saved_tensors
        linear_op = ctx.representation_tree(*matrix_args)
        inv_quad_grad_output = inv_quad_grad_output.unsqueeze(-2)
        neg_solves_times_grad = inv_quad_solves.mul(inv_quad_grad_output).mul(-1)
        matrix_arg_grads = [None] * len(matrix_args)
        if any(ctx.needs_input_grad[2:]):
            matrix_arg_grads = linear_op._bilinear_derivative(neg_solves_times_grad, inv_quad_solves)
        if ctx.needs_input_grad[1]:
            inv_quad_rhs_grad = neg_solves_times_grad.mul(-2)
        else:
            inv_quad_rhs_grad = torch.zeros_like(inv_quad_solves)
        if ctx.is_vector:
            inv_quad_rhs_grad = inv_quad_rhs_grad.squeeze(-1)
        return tuple([None] + [inv_quad_rhs_grad] + list(matrix_arg_grads))
