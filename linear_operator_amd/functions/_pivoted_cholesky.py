"""PivotedCholesky Function (reference: linear_operator/functions/_pivoted_cholesky.py:12-147), forward on the
device through csrc/lo_pivchol.hip (rows generated from the operator descriptor instead of the generic
__getitem__ / gather / scatter chain).  Operators without a descriptor (opaque trees) keep the reference's generic
row access as a per-pivot callback (lo_pivoted_cholesky_cb_f32); nothing is densified."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import kernels as K
from .. import settings


class PivotedCholesky(Function):
    @staticmethod
    def forward(ctx, representation_tree, max_iter, error_tol, *matrix_args):
        matrix = representation_tree(*matrix_args)
        if error_tol is None:
            error_tol = settings.preconditioner_tolerance.value()
        if settings.verbose_linalg.on():
            settings.verbose_linalg.logger.debug(
                f"Running Pivoted Cholesky on a {matrix.shape} RHS for {max_iter} iterations."
            )
        desc = matrix._kernel_descriptor()
        if desc is not None and desc.diag_mode != 0:
            desc = None  # the kernels factor a descriptor WITHOUT its diagonal; a genuine A + D takes the generic path
        if desc is not None:
            L, perm = K.pivoted_cholesky(desc, max_iter, float(error_tol))
        else:
            # no descriptor: the reference's generic accesses -- matrix._diagonal() (:39) and one row per pivot through
            # LinearOperator.__getitem__ (:81) -- feed the same kernels; nothing is densified
            # (float64 operators keep their dtype -- the reference is dtype-generic: lo_pivoted_cholesky_cb_f64)
            diag = matrix._diagonal()
            diag = (diag if diag.dtype == torch.float64 else diag.to(torch.float32)).contiguous()
            L, perm = K.pivoted_cholesky_generic(diag, matrix._get_rows, max_iter, float(error_tol))
        ctx.mark_non_differentiable(perm)
        ctx.representation_tree = representation_tree
        ctx.save_for_backward(perm, *matrix_args)
        return L, perm

    @staticmethod
    def backward(ctx, grad_output, _):  # reference :107-147
        perm, *matrix_args = ctx.saved_tensors
        linear_op = ctx.representation_tree(*matrix_args)
        grads = pivoted_cholesky_vjp(linear_op, perm, grad_output)
        return tuple([None, None, None] + list(grads))


def _dense_root_vjp(r, perm, grad_L, m, factor=None, accumulate_into=None, consume_grad=False):
    """The same pull-back for K = R R^T written out by hand (a dozen passes over [*, N, m] / [*, N, R] data instead of
    the autograd tape of the generic re-expression); only the m x m Cholesky goes through autograd, so the
    triangular / symmetric conventions of its backward are the ones the generic path (and the reference) get.
      pivoted factor = [L11; K21 L11^-T]:  with G = grad_L split into its pivot rows G11 and the others G2, Rest = K21 L11^-T:
      bar L11 = G11 - L11^-T (G2^T Rest),  bar K21 = G2 L11^-1,  bar K11 = chol_backward(bar L11),
      bar Rp = bar Krows Rm,  bar Rm += bar Krows^T Rp   (Krows = Rp Rm^T, Rm = the m pivot rows).
    `factor`: the pivoted-Cholesky factor L [*, N, m] itself when the caller has it (the preconditioner cache does):
    Rest = K21 L11^-T IS its non-pivot rows and L11 its pivot rows, so the N x R x m and N x m x m products that
    rebuild them are skipped.
    `accumulate_into`: a gradient tensor [*, N, R] of the caller's own that the result is to be ADDED to -- the N-sized
    product then runs as one in-place pass over it (kernels.root_apply_add) and the tensor itself is returned; when the
    kernel does not take the operands the result comes back as a new tensor, as without the argument.
    `consume_grad`: grad_L is a temporary of the caller's that may be overwritten (its pivot rows are zeroed in place
    instead of in a copy: one pass over [*, N, m] less)."""
    from .. import kernels as K
    from ..utils.cholesky import psd_safe_cholesky

    R = r.size(-1)
    # Everything stays in the ORIGINAL row order: only the m pivot rows are gathered / scattered (the factor's row
    # perm[i] is row i of the pivoted factor, so "the rows below the pivots" are simply all non-pivot rows).
    piv = perm[..., :m]
    idx_r = piv.unsqueeze(-1).expand(*piv.shape, R)
    idx_m = piv.unsqueeze(-1).expand(*piv.shape, m)
    grad_L = grad_L.contiguous()
    rm = torch.gather(r, -2, idx_r)  # the m pivot rows of the root  [*, m, R]
    if factor is not None:
        rest = factor if factor.is_contiguous() else factor.contiguous()
        k11 = (rm @ rm.mT).detach().requires_grad_(True)  # K[pivots, pivots]  [*, m, m]
    else:
        krows = r @ rm.mT  # K[:, pivots]  [*, N, m]
        k11 = torch.gather(krows, -2, idx_m).detach().clone().requires_grad_(True)
    with torch.enable_grad():
        l11 = psd_safe_cholesky(k11)
    l11d = l11.detach()
    # the N x m triangular solves become GEMMs with the explicit m x m inverse (m <= max_preconditioner_size)
    eye = torch.eye(m, dtype=l11d.dtype, device=l11d.device).expand(*l11d.shape[:-2], m, m)
    l11_inv = torch.linalg.solve_triangular(l11d, eye, upper=False)
    g11 = torch.gather(grad_L, -2, idx_m)
    # gradient of the non-pivot rows (pivot rows zeroed; g11 above holds them)
    g2 = grad_L.scatter_(-2, idx_m, 0.0) if consume_grad else grad_L.scatter(-2, idx_m, 0.0)
    if factor is None:
        rest = krows @ l11_inv.mT  # K21 L11^-T (its pivot rows meet zeros of g2 only)
    lbar = g11 - l11_inv.mT @ (g2.mT @ rest)
    (k11bar,) = torch.autograd.grad(l11, k11, grad_outputs=lbar)
    # bar K[:, pivots] = G2 L11^-1 below the pivots (kb0) and bar K11 on the pivot rows (S); with Krows = R Rm^T
    #   bar R = kb0 Rm + [pivot rows] (S Rm + (kb0 + S)^T R):
    # the N-sized work is ONE product G2 (L11^-1 Rm) and ONE reduction G2^T R, the rest is m x m / m x R algebra
    piv_rows = k11bar @ rm + l11_inv.mT @ (g2.mT @ r) + k11bar.mT @ rm
    w = l11_inv @ rm
    if accumulate_into is not None and tuple(accumulate_into.shape) == tuple(r.shape) and K.root_apply_add(g2, w, accumulate_into):
        return accumulate_into.scatter_add_(-2, idx_r.expand(*accumulate_into.shape[:-2], m, R), piv_rows.expand(*accumulate_into.shape[:-2], m, R))
    r_bar = g2 @ w
    return r_bar.scatter_add_(-2, idx_r, piv_rows)


def pivoted_cholesky_vjp(linear_op, full_permutation, grad_L, generic=False, factor=None, accumulate_into=None,
                         consume_grad=False):
    """Vector-Jacobian product of the pivoted-Cholesky factor L [*batch, N, m] with respect to the tensors that
    represent `linear_op`, the way PivotedCholesky.backward does it (reference :107-147): re-express the factor of
    the SAME pivots as  Pi^T [chol(K_pp); (chol(K_pp)^-1 K_pr)^T]  with differentiable ATen ops on the m pivot rows
    (m x N data, k x k Cholesky: plumbing, like the reference) and back-propagate grad_L through it.
    Returns one gradient (or None) per tensor of linear_op.representation().  Dense roots take the hand-written
    pull-back (_dense_root_vjp) unless `generic` asks for the autograd tape (tests compare the two).
    `accumulate_into` (one entry per tensor of the representation, or None): gradients of the caller's own that the
    results are to be added to; an entry that comes back as the SAME tensor has been updated in place.
    `consume_grad`: grad_L is a temporary that may be overwritten."""
    from ..operators.dense_linear_operator import DenseLinearOperator
    from ..operators.root_linear_operator import RootLinearOperator
    from ..utils.cholesky import psd_safe_cholesky
    from ..utils.permutation import inverse_permutation

    m = grad_L.size(-1)
    perm = full_permutation
    reps = linear_op.representation()
    if not generic and isinstance(linear_op, RootLinearOperator) and len(reps) == 1 and linear_op._dense_root() is reps[0]:
        acc = accumulate_into[0] if accumulate_into else None
        return [_dense_root_vjp(reps[0].detach(), perm, grad_L, m, factor=factor, accumulate_into=acc,
                                consume_grad=consume_grad)]
    inv_perm = inverse_permutation(perm)
    leaves = []
    for t in linear_op.representation():
        leaves.append(t.detach().requires_grad_(True) if t.dtype.is_floating_point else t.detach())
    with torch.enable_grad():
        op = linear_op.representation_tree()(*leaves)
        if isinstance(op, RootLinearOperator) and op._dense_root() is not None:
            r = op._dense_root()
            rp = torch.gather(r, -2, perm.unsqueeze(-1).expand(*perm.shape, r.size(-1)))  # rows in pivot order
            krows = rp @ rp[..., :m, :].mT  # K[perm][:, pivots]  [*batch, N, m]
        elif isinstance(op, DenseLinearOperator):
            kd = op.tensor
            rows = torch.gather(kd, -2, perm.unsqueeze(-1).expand(*perm.shape, kd.size(-1)))
            krows = torch.gather(rows, -1, perm[..., :m].unsqueeze(-2).expand(*perm.shape[:-1], perm.size(-1), m))
        else:
            # any other operator (sums, Kronecker products, non-dense roots ...): the m pivot columns K e_pi through the
            # differentiable Matmul Function (its backward is the operator's _bilinear_derivative), rows put in pivot
            # order afterwards -- the reference reaches the same entries through __getitem__ on the rebuilt operator
            n = op.size(-1)
            onehot = torch.zeros(*perm.shape[:-1], n, m, dtype=grad_L.dtype, device=grad_L.device)
            onehot.scatter_(-2, perm[..., :m].unsqueeze(-2), 1.0)
            kcols = op.matmul(onehot)  # K[:, pivots]  (K symmetric)
            krows = torch.gather(kcols, -2, perm.unsqueeze(-1).expand(*perm.shape, m))
        l11 = psd_safe_cholesky(krows[..., :m, :])
        rest = torch.linalg.solve_triangular(l11, krows[..., m:, :].mT, upper=False).mT
        res_pivoted = torch.cat([l11, rest], dim=-2)
        res = torch.gather(res_pivoted, -2, inv_perm.unsqueeze(-1).expand(*inv_perm.shape, m))
        need = [t for t in leaves if t.requires_grad]
        grads = list(torch.autograd.grad(res, need, grad_outputs=grad_L.contiguous(), allow_unused=True))
    out = []
    for t in leaves:
        out.append(grads.pop(0) if t.requires_grad else None)
    return out
