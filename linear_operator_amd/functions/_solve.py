"""Solve Function: `L A^-1 R` (or `A^-1 R`) with its derivative (reference behaviour: linear_operator/functions/_solve.py:10-131).

Forward: one solve for the stacked block `[L^T | R]`, through the exact Cholesky branch for small operators (ATen
plumbing, BASELINE cfg1) or the device CG with the operator's preconditioner.  Backward: with `S_R = A^-1 R` saved from
the forward and `S_L = A^-1 L^T g` (already available when a left factor was given, one more device solve otherwise)
    d/dR = S_L,   d/dL = g S_R^T,   d/dA = -sym(S_L S_R^T)  -> `_bilinear_derivative` (csrc/lo_bilinear.hip).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings


def _solve(linear_op, rhs):
    """Cholesky for N <= max_cholesky_size (or fast solves switched off), else `linear_op._solve` with the
    preconditioner built from the detached operator (reference :17-22)."""
    small = linear_op.size(-1) <= settings.max_cholesky_size.value()
    if getattr(linear_op, "_has_closed_form_solve", False):
        # operators whose `_solve` is a closed form (Woodbury for LowRankRoot + Diag, per-factor eigendecomposition for
        # Kronecker + constant diagonal) go straight to it -- as the reference's own `solve` of those classes does
        # (low_rank_root_added_diag_linear_operator.py:152, :62-90): no O(N^3) dense Cholesky, no preconditioner build.
        # The closed forms run in liblo_amd (fp32 HIP tensors); a SMALL operator that is not one keeps the exact
        # Cholesky branch below (CPU tensors, fp64: the plumbing case), a large one has no other route and raises there.
        on_device = (linear_op.dtype == torch.float32 and linear_op.device.type == "cuda"
                     and rhs.dtype == torch.float32 and rhs.is_cuda)
        if on_device or not small:
            return linear_op._solve(rhs)
    if small or settings.fast_computations.solves.off():
        return linear_op.cholesky()._cholesky_solve(rhs)
    with torch.no_grad():
        # (detached copy, as the reference does -- unless nothing of the operator takes part in autograd: rebuilding
        # the operator tree costs ~10 us of host time per solve)
        base = linear_op.detach() if linear_op.requires_grad else linear_op
        preconditioner = base._solve_preconditioner()
    return linear_op._solve(rhs, preconditioner)


def _symmetric_operator_grads(linear_op, s_left, s_right):
    """Gradients w.r.t. the operator's tensors of -<S_L, dA S_R>, symmetrised: both orders stacked along the columns
    with weight -1/2 each (the operator is symmetric, its representation may not be)."""
    left = torch.cat((s_left, s_right), dim=-1)
    right = torch.cat((s_right, s_left), dim=-1) * -0.5
    return linear_op._bilinear_derivative(left, right)


class Solve(Function):
    @staticmethod
    def forward(ctx, representation_tree, has_left, *args):
        n_lead = 2 if has_left else 1
        lead, matrix_args = args[:n_lead], args[n_lead:]
        left = lead[0] if has_left else None
        right = lead[-1]
        ctx.representation_tree, ctx.has_left = representation_tree, has_left
        ctx.is_vector = right.dim() == 1
        cols = right.unsqueeze(-1) if ctx.is_vector else right
        linear_op = representation_tree(*matrix_args)
        if has_left:
            n_left = left.size(-2)
            solves = _solve(linear_op, torch.cat((left.mT, cols), dim=-1))  # [A^-1 L^T | A^-1 R] in one call
            out = left @ solves[..., n_left:]
            ctx.save_for_backward(solves, left, right, *matrix_args)
        else:
            solves = out = _solve(linear_op, cols)
            ctx.save_for_backward(solves, right, *matrix_args)
        return out.squeeze(-1) if ctx.is_vector else out

    @staticmethod
    def backward(ctx, grad_output):
        saved = ctx.saved_tensors
        n_head = 3 if ctx.has_left else 2
        matrix_args = saved[n_head:]
        n_fixed = 2  # representation_tree, has_left
        none_head = (None,) * n_fixed
        if not any(ctx.needs_input_grad):
            return none_head + (None,) * (len(saved) - 1)
        g = grad_output.unsqueeze(-1) if ctx.is_vector else grad_output
        if ctx.has_left:
            solves, left, _ = saved[:3]
            n_left = left.size(-2)
            s_right = solves[..., n_left:]
            s_left = solves[..., :n_left] @ g  # A^-1 L^T g without another solve
            need_left, need_right = ctx.needs_input_grad[n_fixed], ctx.needs_input_grad[n_fixed + 1]
            need_args = any(ctx.needs_input_grad[n_fixed + 2:])
        else:
            s_right = saved[0]
            s_left = Solve.apply(ctx.representation_tree, False, g.contiguous(), *matrix_args)  # A^-1 g on the device
            need_left, need_right = False, ctx.needs_input_grad[n_fixed]
            need_args = any(ctx.needs_input_grad[n_fixed + 1:])
        arg_grads = (None,) * len(matrix_args)
        if need_args:
            arg_grads = tuple(_symmetric_operator_grads(ctx.representation_tree(*matrix_args), s_left, s_right))
        right_grad = None
        if need_right:
            right_grad = s_left.squeeze(-1) if ctx.is_vector else s_left
        if not ctx.has_left:
            return none_head + (right_grad,) + arg_grads
        left_grad = g @ s_right.mT if need_left else None
        return none_head + (left_grad, right_grad) + arg_grads
