"""minres with the reference's signature and semantics (linear_operator/utils/minres.py:10-207), executed by
liblo_amd's shifted-MINRES engine (csrc/lo_minres.hip).  Closure handling is the one of utils/linear_cg.py: operator
`_matmul`s that lower to a kernel descriptor run entirely on the device, other callables are called back for the
product only; the Woodbury preconditioner closure of AddedDiagLinearOperator is applied natively.  HIP tensors only --
there is no CPU implementation; float64 operands (the reference's own test recipes) take csrc/lo_minres_f64.hip."""
from __future__ import annotations

import torch

from .. import kernels as K
from .. import settings
from .linear_cg import _lower_matmul_closure


def minres(matmul_closure, rhs, eps=1e-25, shifts=None, value=None, max_iter=None, preconditioner=None):
    r"""Solutions of :math:`(\alpha K + \sigma_q I) x_q = b` for all shifts :math:`\sigma_q` at once (``value`` is
    :math:`\alpha`).  Returns ``[Q, *batch, N, c]`` -- without the leading dimension when there is a single shift."""
    if not torch.is_tensor(matmul_closure) and not callable(matmul_closure):
        raise RuntimeError("matmul_closure must be a tensor, or a callable object!")
    if shifts is None:  # :43-44
        shifts = torch.tensor(0.0, dtype=rhs.dtype, device=rhs.device)
    squeeze = rhs.dim() == 1  # :47-50
    if squeeze:
        rhs = rhs.unsqueeze(-1)
    if max_iter is None:  # :58-60
        max_iter = settings.max_cg_iterations.value()
    max_iter = min(max_iter, rhs.size(-2) + 1)

    # batch shape of the products (the reference learns it from one product, :64)
    owner = getattr(matmul_closure, "__self__", None)
    if torch.is_tensor(matmul_closure):
        op_batch = matmul_closure.shape[:-2]
    elif owner is not None and hasattr(owner, "batch_shape"):
        op_batch = owner.batch_shape
    else:
        op_batch = matmul_closure(rhs).shape[:-2]
    batch_shape = torch.broadcast_shapes(op_batch, rhs.shape[:-2])
    rhs_b = rhs.expand(*batch_shape, *rhs.shape[-2:]).contiguous()

    sh = shifts.reshape(1) if shifts.dim() == 0 else shifts
    if rhs.dtype == torch.float64:
        # the reference's fp64 recipes (test/utils/test_minres.py): dense tensors on the library's fp64 kernel, any
        # other closure / preconditioner called back (csrc/lo_minres_f64.hip)
        dense = matmul_closure if torch.is_tensor(matmul_closure) and matmul_closure.dim() >= 2 else None
        res = K.minres_solve_f64(
            dense, rhs_b, sh, value=value,
            matvec_closure=None if dense is not None else (
                matmul_closure.matmul if torch.is_tensor(matmul_closure) else matmul_closure),
            precond_closure=preconditioner, max_iter=max_iter, tolerance=float(settings.minres_tolerance.value()),
            eps=float(eps),
        )
        solution = res.x
        if squeeze:
            solution = solution.squeeze(-1)
        if shifts.numel() == 1:
            solution = solution.squeeze(0)
        return solution

    desc = _lower_matmul_closure(matmul_closure, batch_shape)
    closure = None
    if desc is None:
        closure = matmul_closure.matmul if torch.is_tensor(matmul_closure) else matmul_closure
    woodbury, precond_closure = None, None
    if preconditioner is not None:
        woodbury = getattr(preconditioner, "woodbury", None)
        if woodbury is not None and int(woodbury.dinv.shape[0]) != max(1, batch_shape.numel()):
            woodbury = None
        if woodbury is None:
            precond_closure = preconditioner
    if settings.verbose_linalg.on():
        settings.verbose_linalg.logger.debug(
            f"Running MINRES on a {rhs.shape} RHS for {max_iter} iterations "
            f"(tol={settings.minres_tolerance.value()}). Output: {(sh.shape[0], *rhs_b.shape)}."
        )
    res = K.minres_solve(
        desc, rhs_b, sh, value=value, precond=woodbury, matvec_closure=closure, precond_closure=precond_closure,
        max_iter=max_iter, tolerance=float(settings.minres_tolerance.value()), eps=float(eps),
    )
    solution = res.x
    if squeeze:  # :203-206
        solution = solution.squeeze(-1)
    if shifts.numel() == 1:  # :208-210
        solution = solution.squeeze(0)
    return solution


__all__ = ["minres"]
