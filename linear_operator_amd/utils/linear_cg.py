"""linear_cg with the reference's signature and semantics (linear_operator/utils/linear_cg.py:98-359), executed
by liblo_amd's device-side CG engine (csrc/lo_cg.hip).  This module attribute is THE solver seam of the
reference: `LinearOperator._solve` resolves `utils.linear_cg` at call time (operators/_linear_operator.py:796), and
the reference's tests wrap it with mock.patch -- both keep working here.

How the closure is executed:
  * `matmul_closure` is a torch.Tensor, or the bound `_matmul` of an operator whose tree lowers to a kernel
    descriptor (AddedDiag(LowRankRoot|Dense|Kron, Diag) ...): the whole loop -- matvec included -- runs on the
    device with no Python in it;
  * any other callable: the engine calls back into Python for the product only (one launch group per
    iteration, vector updates / inner products / stopping rule stay in the kernels);
  * `preconditioner`: the Woodbury closure AddedDiagLinearOperator._preconditioner returns is recognised and
    applied natively; any other callable is called back.
There is no CPU implementation: tensors must be HIP tensors, fp32 (every engine) or fp64 (csrc/lo_cg_f64.hip: dense
tensors and closures, what the reference's own test/utils/test_linear_cg.py runs).
"""
from __future__ import annotations

import threading
import warnings

import torch

from .. import settings
from .. import kernels as K
from .warnings import NumericalWarning


def _active_stop_reduce():
    """The all-reduce hook of `distributed.global_stopping_rule(...)` when one is active (batch-sharded solves that
    keep the reference's batch-GLOBAL stopping rule, linear_cg.py:302-308), else None."""
    from .. import distributed

    return distributed.active_stop_reduce()


def _default_preconditioner(x):
    return x.clone()


# ---- the one-launch solve is speculative: a launch whose batch-global decisions do not hold (a member stops pivoting
# early, the tolerance is not met at the 11-iteration floor, lo_amd.h: LO_FUSED_EARLY_STOP / _CONTINUE) is discarded and the
# three-launch path redoes everything.  A system that missed is likely to miss again (the same operator in the next
# solve, the next step of a training loop): remember the miss and skip the speculation for the next few solves of the
# same operator / settings, then try again.
_FUSED_MISSES: dict = {}
_FUSED_MISSES_LOCK = threading.Lock()
_FUSED_SKIP_AFTER_MISS = 8


def _fused_key(desc, rhs, rank, pc_tol, tolerance, n_iter):
    """Signature of a solve whose one-launch attempt missed.  Address AND version counter of the leaves: a freed
    operator's address reused by another operator, or the same tensor after an in-place update, is a new key."""
    d = getattr(desc, "d", None)
    return (desc.A0.data_ptr(), desc.A0._version, 0 if d is None else d.data_ptr(), 0 if d is None else d._version,
            desc.B, desc.N, desc.R, int(rhs.shape[-1]), int(rank), float(pc_tol), float(tolerance), int(n_iter))


def _fused_worth_trying(key) -> bool:
    with _FUSED_MISSES_LOCK:
        left = _FUSED_MISSES.get(key)
        if left is None:
            return True
        if left <= 0:
            del _FUSED_MISSES[key]  # (a retry: conditions may have changed)
            return True
        _FUSED_MISSES[key] = left - 1
        return False


def _fused_note(key, hit: bool):
    with _FUSED_MISSES_LOCK:
        if hit:
            _FUSED_MISSES.pop(key, None)
            return
        if len(_FUSED_MISSES) >= 64:
            _FUSED_MISSES.pop(next(iter(_FUSED_MISSES)))
        _FUSED_MISSES[key] = _FUSED_SKIP_AFTER_MISS


def _lower_matmul_closure(matmul_closure, batch_shape):
    """Return an OperatorDescriptor for closures we can run natively, else None."""
    if torch.is_tensor(matmul_closure):
        M = matmul_closure
        if M.dim() < 2 or M.shape[-1] != M.shape[-2]:
            return None
        return K.dense_diag_descriptor(M.expand(*batch_shape, *M.shape[-2:]), None)
    owner = getattr(matmul_closure, "__self__", None)
    if owner is not None and getattr(matmul_closure, "__name__", "") == "_matmul" and hasattr(owner, "_kernel_descriptor"):
        if type(owner)._matmul is not getattr(matmul_closure, "__func__", None):
            return None  # instance-level override / mock
        return owner._kernel_descriptor(batch_shape)
    return None


def linear_cg(
    matmul_closure,
    rhs,
    n_tridiag=0,
    tolerance=None,
    eps=1e-10,
    stop_updating_after=1e-10,
    max_iter=None,
    max_tridiag_iter=None,
    initial_guess=None,
    preconditioner=None,
):
    """Solve `lhs result = rhs` for a (batch of) symmetric positive definite operators.

    Returns `result`, or `(result, tridiags)` when n_tridiag > 0 (tridiags: [n_tridiag, *batch, T, T]).
    """
    is_vector = rhs.ndimension() == 1  # linear_cg.py:134-136
    if is_vector:
        rhs = rhs.unsqueeze(-1)
    if max_iter is None:
        max_iter = settings.max_cg_iterations.value()
    if max_tridiag_iter is None:
        max_tridiag_iter = settings.max_lanczos_quadrature_iterations.value()
    if initial_guess is not None and initial_guess.ndimension() == 1:  # :147-149
        is_vector = True
        initial_guess = initial_guess.unsqueeze(-1)
    if tolerance is None:
        tolerance = settings.cg_tolerance.value()
    if max_tridiag_iter > max_iter:  # :159-160 (raised even when n_tridiag == 0)
        raise RuntimeError("Getting a tridiagonalization larger than the number of CG iterations run is not possible!")
    if not torch.is_tensor(matmul_closure) and not callable(matmul_closure):  # :163-166
        raise RuntimeError("matmul_closure must be a tensor, or a callable object!")

    num_rows = rhs.size(-2)
    n_iter = min(max_iter, num_rows) if settings.terminate_cg_by_size.on() else max_iter  # :170
    n_tridiag_iter = min(max_tridiag_iter, num_rows)  # :171
    # The reference broadcasts a batched operator against an unbatched right-hand side inside the closure's matmul
    # (residual = rhs - matmul_closure(x0), :186); the kernels want the broadcast made explicit.
    batch_shape = rhs.shape[:-2]
    if torch.is_tensor(matmul_closure):
        op_batch = matmul_closure.shape[:-2]
    else:
        op_batch = getattr(getattr(matmul_closure, "__self__", None), "batch_shape", torch.Size())
    if op_batch == batch_shape:
        full_batch = batch_shape
    else:
        try:
            full_batch = torch.broadcast_shapes(batch_shape, op_batch)
        except RuntimeError:
            full_batch = batch_shape
    if full_batch != batch_shape:
        rhs = rhs.expand(*full_batch, *rhs.shape[-2:])
        if initial_guess is not None:
            initial_guess = initial_guess.expand(*full_batch, *initial_guess.shape[-2:])
        batch_shape = full_batch

    if settings.verbose_linalg.on():
        settings.verbose_linalg.logger.debug(
            f"Running CG on a {rhs.shape} RHS for {n_iter} iterations (tol={tolerance}). Output: {rhs.shape}."
        )

    if rhs.dtype == torch.float64:
        # the reference's fp64 recipes (test/utils/test_linear_cg.py): a dense tensor is multiplied by the library's
        # fp64 kernel, any other closure (and any preconditioner) is called back; no Woodbury / resident engines
        if _active_stop_reduce() is not None:
            raise NotImplementedError("the batch-global stopping rule over ranks is fp32 only")
        dense = matmul_closure if torch.is_tensor(matmul_closure) and matmul_closure.dim() >= 2 else None
        res = K.cg_solve_f64(
            dense, None, rhs, x0=initial_guess,
            matvec_closure=None if dense is not None else (
                matmul_closure.matmul if torch.is_tensor(matmul_closure) else matmul_closure),
            precond_closure=preconditioner, n_tridiag=n_tridiag, max_iter=n_iter, max_tridiag_iter=n_tridiag_iter,
            tolerance=float(tolerance), eps=float(eps), stop_updating_after=float(stop_updating_after),
            floor_max_iter=max_iter,
        )
    else:
        lazy = preconditioner if getattr(preconditioner, "lazy_fused", False) and preconditioner.pending else None
        if lazy is not None and lazy.owns(matmul_closure, batch_shape):
            desc = lazy.desc  # (the closure is the `_matmul` of the operator the lazy closure was lowered from)
        else:
            desc = _lower_matmul_closure(matmul_closure, batch_shape)
        closure = None
        if desc is None:
            closure = matmul_closure.matmul if torch.is_tensor(matmul_closure) else matmul_closure
        res = None
        if lazy is not None:
            # AddedDiagLinearOperator._solve_preconditioner deferred the factorisation: when this call is what the
            # one-launch kernel computes (the closure's own operator, no tridiagonals, zero initial guess, the rule of
            # this process only) pivoted Cholesky, root-form preconditioner and CG run in ONE resident launch
            if (n_tridiag == 0 and initial_guess is None and _active_stop_reduce() is None
                    and (desc is lazy.desc or lazy.same_operator(desc)) and rhs.is_cuda):
                key = _fused_key(desc, rhs, lazy.rank, lazy.tol, tolerance, n_iter)
                if _fused_worth_trying(key):
                    fused = K.solve_fused(desc, rhs, lazy.rank, lazy.tol, max_iter=n_iter, tolerance=float(tolerance),
                                          eps=float(eps), stop_updating_after=float(stop_updating_after),
                                          floor_max_iter=max_iter)
                    _fused_note(key, fused is not None)
                    if fused is not None:
                        lazy.adopt(fused.precond)
                        res = fused.cg
            if res is None:
                preconditioner = lazy.materialize()  # the ordinary three-launch build (or None: NaN in the factor)
        woodbury, precond_closure = None, None
        if preconditioner is not None and res is None:
            woodbury = getattr(preconditioner, "woodbury", None)
            if woodbury is not None and int(woodbury.dinv.shape[0]) != max(1, batch_shape.numel()):
                woodbury = None
            if woodbury is None:
                precond_closure = preconditioner

        res = res if res is not None else K.cg_solve(
            desc, rhs, x0=initial_guess, precond=woodbury, matvec_closure=closure, precond_closure=precond_closure,
            n_tridiag=n_tridiag, max_iter=n_iter, max_tridiag_iter=n_tridiag_iter, tolerance=float(tolerance),
            eps=float(eps), stop_updating_after=float(stop_updating_after), floor_max_iter=max_iter,
            stop_reduce=_active_stop_reduce(),
        )
    if res.nan_detected:  # :199-200
        raise RuntimeError("NaNs encountered when trying to perform matrix-vector multiplication")
    if not res.tolerance_reached and res.iterations > 0:  # :337-347
        warnings.warn(
            "CG terminated in {} iterations with average residual norm {}"
            " which is larger than the tolerance of {} specified by"
            " linear_operator.settings.cg_tolerance."
            " If performance is affected, consider raising the maximum number of CG iterations by running code in"
            " a linear_operator.settings.max_cg_iterations(value) context.".format(
                res.iterations, res.mean_residual, tolerance
            ),
            NumericalWarning,
        )
    result = res.x
    if is_vector:
        result = result.squeeze(-1)
    if n_tridiag:
        t = res.t_mat  # [n_tridiag, B, T', T']
        t = t.reshape(n_tridiag, *batch_shape, t.shape[-2], t.shape[-1])
        return result, t
    return result
