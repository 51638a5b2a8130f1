"""Matmul Function (reference: linear_operator/functions/_matmul.py:9-66), forward and backward."""
from __future__ import annotations

from torch.autograd import Function


class Matmul(Function):
    @staticmethod
    def forward(ctx, representation_tree, rhs, *matrix_args):
        ctx.representation_tree = representation_tree
        orig_rhs = rhs
        linear_op = representation_tree(*matrix_args)
        is_vec = rhs.ndimension() == 1
        if is_vec:
            rhs = rhs.unsqueeze(-1)
        res = linear_op._matmul(rhs)
        ctx.save_for_backward(orig_rhs, *matrix_args)
        return res.squeeze(-1) if is_vec else res

    @staticmethod
    def backward(ctx, grad_output):  # reference :37-66
        rhs = ctx.saved_tensors[0]
        matrix_args = ctx.saved_tensors[1:]
        rhs_shape = rhs.shape
        rhs_grad = None
        arg_grads = [None] * len(matrix_args)
        if any(ctx.needs_input_grad[2:]):
            rhs_m = rhs.unsqueeze(-1) if rhs.ndimension() == 1 else rhs
            grad_m = grad_output.unsqueeze(-1) if grad_output.ndimension() == 1 else grad_output
            arg_grads = ctx.representation_tree(*matrix_args)._bilinear_derivative(grad_m, rhs_m)
        if ctx.needs_input_grad[1]:
            linear_op = ctx.representation_tree(*matrix_args)
            if grad_output.dim() == 1:
                rhs_grad = linear_op._t_matmul(grad_output.unsqueeze(-1)).squeeze(-1)
            else:
                rhs_grad = linear_op._t_matmul(grad_output.contiguous())
            if rhs_grad.dim() > len(rhs_shape):  # broadcasting
                rhs_grad = rhs_grad.reshape(-1, *rhs_shape).sum(0)
        return tuple([None] + [rhs_grad] + list(arg_grads))
