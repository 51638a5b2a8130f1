"""InvQuad Function (reference: linear_operator/functions/_inv_quad.py:10-61): diag(R^T A^-1 R)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings
from ._common import not_yet


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
        linear_op = representation_tree(*matrix_args)
        is_vector = inv_quad_rhs.ndimension() == 1
        if is_vector:
            inv_quad_rhs = inv_quad_rhs.unsqueeze(-1)
        solves = _solve(linear_op, inv_quad_rhs)
        return (solves * inv_quad_rhs).sum(-2)

    @staticmethod
    def backward(ctx, grad_output):
        not_yet("InvQuad")
