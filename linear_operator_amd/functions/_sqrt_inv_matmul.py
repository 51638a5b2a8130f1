"""SqrtInvMatmul Function (reference: linear_operator/functions/_sqrt_inv_matmul.py:10-140): A^{-1/2} rhs or
lhs A^{-1/2} rhs (+ the inverse quadratic form of lhs) by contour integral quadrature -- shifted MINRES on the device
(csrc/lo_minres.hip) -- forward and backward (`_bilinear_derivative` kernels, csrc/lo_bilinear.hip)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings, utils


class SqrtInvMatmul(Function):
    @staticmethod
    def forward(ctx, representation_tree, rhs, lhs, *matrix_args):
        ctx.representation_tree = representation_tree
        ctx.linear_op = representation_tree(*matrix_args)
        nq = settings.num_contour_quadrature.value()
        if lhs is not None:  # :23-35
            terms = torch.cat([rhs, lhs.mT], dim=-1)
            solves, weights, no_shift_solves, shifts = utils.contour_integral_quad(
                ctx.linear_op, terms, inverse=True, num_contour_quadrature=nq)
            rhs_solves, lhs_solves = solves.split([rhs.size(-1), lhs.size(-2)], dim=-1)
            lhs_no_shift_solves = no_shift_solves[..., -lhs.size(-2):]
            sqrt_inv_matmul_res = lhs @ (rhs_solves * weights).sum(0)
            inv_quad_res = (lhs_no_shift_solves.mT * lhs).sum(dim=-1).mul_(-1)
        else:  # :36-47
            rhs_solves, weights, _, shifts = utils.contour_integral_quad(
                ctx.linear_op, rhs, inverse=True, num_contour_quadrature=nq)
            sqrt_inv_matmul_res = (rhs_solves * weights).sum(0)
            lhs_solves = None
            lhs_no_shift_solves = None
            inv_quad_res = torch.zeros(ctx.linear_op.batch_shape, dtype=rhs.dtype, device=rhs.device)
        ctx.save_for_backward(rhs, lhs, rhs_solves, lhs_solves, weights, shifts, lhs_no_shift_solves, *matrix_args)
        return sqrt_inv_matmul_res, inv_quad_res

    @staticmethod
    def backward(ctx, sqrt_inv_matmul_grad, inv_quad_grad):  # :53-140
        rhs, lhs, rhs_solves, lhs_solves, weights, shifts, lhs_no_shift_solves, *matrix_args = ctx.saved_tensors
        rhs_grad = None
        lhs_grad = None
        if lhs is not None:
            weighted_rhs_solves_mul_grad = rhs_solves.mul(weights) @ sqrt_inv_matmul_grad.mT
            neg_inv_quad_solves_mul_grad = lhs_no_shift_solves.mul(inv_quad_grad.unsqueeze(-2)).mul(-1)
            if ctx.needs_input_grad[2]:
                lhs_grad = weighted_rhs_solves_mul_grad.mT.sum(0)
                lhs_grad = lhs_grad.add(neg_inv_quad_solves_mul_grad.mT, alpha=2)
            if ctx.needs_input_grad[1]:
                rhs_grad = (lhs_solves @ sqrt_inv_matmul_grad).mul(weights).sum(0)
            terms1 = torch.cat([lhs_no_shift_solves.unsqueeze(0), lhs_solves], 0)
            terms2 = torch.cat([neg_inv_quad_solves_mul_grad.unsqueeze(0), weighted_rhs_solves_mul_grad], 0)
        else:
            grad_solves, _, _, _ = utils.contour_integral_quad(
                ctx.linear_op, sqrt_inv_matmul_grad.contiguous(), inverse=True, weights=weights, shifts=shifts,
                num_contour_quadrature=settings.num_contour_quadrature.value())
            grad_solves_mul_weights = grad_solves.mul(weights)
            if ctx.needs_input_grad[1]:
                rhs_grad = grad_solves_mul_weights.sum(0)
            terms1 = grad_solves_mul_weights
            terms2 = rhs_solves
        matrix_arg_grads = [None] * len(matrix_args)
        if any(ctx.needs_input_grad[3:]):
            # the quadrature dimension rides in the column dimension of the contraction: sum_q sum_d u v^T
            def fold(t):
                return t.movedim(0, -2).reshape(*t.shape[1:-1], t.shape[0] * t.shape[-1])

            left = torch.cat([fold(terms1), fold(terms2)], -1)
            right = torch.cat([fold(terms2), fold(terms1)], -1).mul_(0.5)
            matrix_arg_grads = ctx.linear_op._bilinear_derivative(left.contiguous(), right.contiguous())
        return (None, rhs_grad, lhs_grad, *matrix_arg_grads)


__all__ = ["SqrtInvMatmul"]
