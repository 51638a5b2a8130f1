"""Solve Function + the Cholesky / CG dispatch, forward and backward (reference: linear_operator/functions/_solve.py:10-131).
Backward = one more solve for A^-1 grad and the `_bilinear_derivative` contraction (csrc/lo_bilinear.hip)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings


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
        orig_right_tensor = right_tensor
        linear_op = representation_tree(*matrix_args)
        ctx.representation_tree = representation_tree
        ctx.has_left = has_left
        ctx.is_vector = right_tensor.ndimension() == 1
        if ctx.is_vector:
            right_tensor = right_tensor.unsqueeze(-1)
        if has_left:  # one solve for [L^T | R], then L @ A^-1 R (reference :49-53)
            rhs = torch.cat([left_tensor.mT, right_tensor], -1)
            solves = _solve(linear_op, rhs)
            res = left_tensor @ solves[..., left_tensor.size(-2):]
            ctx.save_for_backward(solves, left_tensor, orig_right_tensor, *matrix_args)  # :60-66
        else:
            solves = _solve(linear_op, right_tensor)
            res = solves
            ctx.save_for_backward(solves, orig_right_tensor, *matrix_args)
        return res.squeeze(-1) if ctx.is_vector else res

    @staticmethod
    def backward(ctx, grad_output):  # reference :70-131
        if ctx.has_left:
            solves, left_tensor, right_tensor, *matrix_args = ctx.saved_tensors
            left_solves = solves[..., : left_tensor.size(-2)]
            right_solves = solves[..., left_tensor.size(-2):]
        else:
            right_solves, right_tensor, *matrix_args = ctx.saved_tensors
        linear_op = ctx.representation_tree(*matrix_args)
        arg_grads = [None] * len(matrix_args)
        left_grad = None
        right_grad = None
        if not any(ctx.needs_input_grad):
            return tuple([None, None] + ([None] if ctx.has_left else []) + [None] + arg_grads)
        if ctx.is_vector:
            right_tensor = right_tensor.unsqueeze(-1)
            grad_output = grad_output.unsqueeze(-1)
        if not ctx.has_left:
            left_solves = Solve.apply(ctx.representation_tree, False, grad_output.contiguous(), *matrix_args)  # A^-1 g
            if any(ctx.needs_input_grad[3:]):
                # symmetric in (left, right): concatenate both orders and halve (:101-107)
                arg_grads = linear_op._bilinear_derivative(
                    torch.cat([left_solves, right_solves], -1),
                    torch.cat([right_solves, left_solves], -1).mul(-0.5),
                )
            if ctx.needs_input_grad[2]:
                right_grad = left_solves
                if ctx.is_vector:
                    right_grad = right_grad.squeeze(-1)
            return tuple([None, None] + [right_grad] + list(arg_grads))
        left_solves = left_solves @ grad_output
        if ctx.needs_input_grad[2]:
            left_grad = grad_output @ right_solves.mT
        if any(ctx.needs_input_grad[4:]):
            arg_grads = linear_op._bilinear_derivative(
                torch.cat([left_solves, right_solves], -1),
                torch.cat([right_solves, left_solves], -1).mul(-0.5),
            )
        if ctx.needs_input_grad[3]:
            right_grad = left_solves
            if ctx.is_vector:
                right_grad = right_grad.squeeze(-1)
        return tuple([None, None] + [left_grad, right_grad] + list(arg_grads))
