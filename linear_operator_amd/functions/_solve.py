"""Solve Function + the Cholesky / CG dispatch (reference: linear_operator/functions/_solve.py:10-68)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings
from ._common import not_yet


def _solve(linear_op, rhs):
    """N <= max_cholesky_size or fast solves off -> exact Cholesky (ATen plumbing, cfg1);
    else preconditioned CG on the device (reference :17-22)."""
    if settings.fast_computations.solves.off() or linear_op.size(-1) <= settings.max_cholesky_size.value():
        return linear_op.cholesky()._cholesky_solve(rhs)
    with torch.no_grad():
        preconditioner = linear_op.detach()._solve_preconditioner()
    return linear_op._solve(rhs, preconditioner)


class Solve(Function):
    @staticmethod
    def forward(ctx, representation_tree, has_left, *args):
        if has_left:
            left_tensor, right_tensor, *matrix_args = args
        else:
            left_tensor = None
            right_tensor, *matrix_args = args
        linear_op = representation_tree(*matrix_args)
        is_vector = right_tensor.ndimension() == 1
        if is_vector:
            right_tensor = right_tensor.unsqueeze(-1)
        if has_left:  # one solve for [L^T | R], then L @ A^-1 R (reference :49-53)
            rhs = torch.cat([left_tensor.mT, right_tensor], -1)
            solves = _solve(linear_op, rhs)
            res = left_tensor @ solves[..., left_tensor.size(-2):]
        else:
            res = _solve(linear_op, right_tensor)
        return res.squeeze(-1) if is_vector else res

    @staticmethod
    def backward(ctx, grad_output):
        not_yet("Solve")
