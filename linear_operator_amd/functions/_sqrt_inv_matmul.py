"""SqrtInvMatmul Function: `A^{-1/2} R` or `L A^{-1/2} R` (plus the inverse quadratic form of L) by contour integral
quadrature -- shifted MINRES on the device (csrc/lo_minres.hip) -- with its derivative (reference behaviour:
linear_operator/functions/_sqrt_inv_matmul.py:10-140).

With the quadrature `A^{-1/2} ~= sum_q w_q (A + t_q I)^{-1}` (utils/contour_integral_quad.py) and the shifted solves
`S_q(X) = (A + t_q I)^{-1} X`:

    forward    H = sum_q w_q S_q(R);   out = H  or  L H;   quad_b = -sum_rows (A^{-1} L^T) o L^T   (as the reference signs it)
    d out/dR   sum_q w_q S_q(L^T G)          (no L: sum_q w_q S_q(G), one more quadrature call with the SAME shifts)
    d out/dL   (sum_q w_q S_q(R) G^T)^T  + 2 (quad term)
    d out/dA   every term has the shape -<U, dA V> with U, V shifted solves: the pairs (U_q, V_q) of all nodes are
               folded into the column dimension and handed to `_bilinear_derivative` ONCE, symmetrised
               (functions/_solve._symmetric_operator_grads: the helper Solve / InvQuad use for the same purpose).
One quadrature call serves both blocks when a left factor is given: the columns `[R | L^T]` are solved together and
sliced, there is no separate code path per argument pattern.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings, utils
from ._solve import _symmetric_operator_grads


def _nodes_into_columns(t):
    """[Q, *batch, N, c] -> [*batch, N, Q c]: the quadrature nodes become extra columns of one contraction."""
    return t.movedim(0, -2).reshape(*t.shape[1:-1], t.shape[0] * t.shape[-1]).contiguous()


class SqrtInvMatmul(Function):
    @staticmethod
    def forward(ctx, representation_tree, rhs, lhs, *matrix_args):
        linear_op = representation_tree(*matrix_args)
        ctx.linear_op, ctx.has_lhs = linear_op, lhs is not None
        n_rhs = rhs.size(-1)
        block = torch.cat((rhs, lhs.mT), dim=-1) if ctx.has_lhs else rhs
        solves, weights, plain, shifts = utils.contour_integral_quad(
            linear_op, block, inverse=True, num_contour_quadrature=settings.num_contour_quadrature.value())
        rhs_solves = solves[..., :n_rhs]  # S_q(R) for every node q
        half = (rhs_solves * weights).sum(0)  # A^{-1/2} R
        if ctx.has_lhs:
            lhs_solves, lhs_plain = solves[..., n_rhs:], plain[..., n_rhs:]
            out = lhs @ half
            quad = (lhs_plain.mT * lhs).sum(dim=-1).neg_()
        else:
            lhs_solves = lhs_plain = None
            out = half
            quad = torch.zeros(linear_op.batch_shape, dtype=rhs.dtype, device=rhs.device)
        ctx.save_for_backward(rhs, lhs, rhs_solves, lhs_solves, lhs_plain, weights, shifts, *matrix_args)
        return out, quad

    @staticmethod
    def backward(ctx, grad_out, grad_quad):
        rhs, lhs, rhs_solves, lhs_solves, lhs_plain, weights, shifts, *matrix_args = ctx.saved_tensors
        want_rhs, want_lhs, want_args = ctx.needs_input_grad[1], ctx.needs_input_grad[2], any(ctx.needs_input_grad[3:])
        d_rhs = d_lhs = None
        if ctx.has_lhs:
            through_out = (rhs_solves * weights) @ grad_out.mT  # w_q S_q(R) G^T, one block per node
            through_quad = lhs_plain * grad_quad.unsqueeze(-2).neg()
            if want_lhs:
                d_lhs = through_out.sum(0).mT + 2.0 * through_quad.mT
            if want_rhs:
                d_rhs = ((lhs_solves @ grad_out) * weights).sum(0)
            # operator pairs: (A^-1 L^T, quad term) for the quadratic form, (S_q(L^T), w_q S_q(R) G^T) for every node
            u_side = torch.cat((lhs_plain.unsqueeze(0), lhs_solves), dim=0)
            v_side = torch.cat((through_quad.unsqueeze(0), through_out), dim=0)
        else:
            g_solves, _, _, _ = utils.contour_integral_quad(
                ctx.linear_op, grad_out.contiguous(), inverse=True, weights=weights, shifts=shifts,
                num_contour_quadrature=settings.num_contour_quadrature.value())
            u_side = g_solves * weights  # w_q S_q(G)
            v_side = rhs_solves
            if want_rhs:
                d_rhs = u_side.sum(0)
        d_args = (None,) * len(matrix_args)
        if want_args:
            # sum_q <U_q, dA V_q>, symmetrised; the shared helper computes -sym(<left, dA right>), hence the sign
            d_args = tuple(_symmetric_operator_grads(ctx.linear_op, _nodes_into_columns(u_side),
                                                     _nodes_into_columns(v_side).neg()))  # (out of place: the reshape may be a view of a saved tensor)
        return (None, d_rhs, d_lhs, *d_args)


__all__ = ["SqrtInvMatmul"]
