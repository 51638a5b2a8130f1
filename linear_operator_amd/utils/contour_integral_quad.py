"""contour_integral_quad with the reference's signature (linear_operator/utils/contour_integral_quad.py:14-156):
K^{1/2} b or K^{-1/2} b as a quadrature over shifted solves,  K^{-1/2} = (2/pi) int_0^inf (K + t^2 I)^{-1} dt.
Device work: the Lanczos estimate of the spectrum's ends (linear_cg with one tridiagonal column: the operator-resident
CG kernel when the operator lowers to it) and ALL shifted solves in one shifted-MINRES run (csrc/lo_minres.hip).
The quadrature nodes / weights come from Jacobi elliptic functions evaluated with scipy on the host for each batch
member, exactly as in the reference (:100-126) -- a few dozen scalars."""
from __future__ import annotations

import math
import warnings

import torch

from .. import settings
from .linear_cg import linear_cg
from .minres import minres
from .warnings import NumericalWarning


def contour_integral_quad(linear_op, rhs, inverse=False, weights=None, shifts=None, max_lanczos_iter=20,
                          num_contour_quadrature=None, shift_offset=0):
    """Returns (solves [Q, *batch, N, c], weights [Q, *batch, 1, 1], no_shift_solves [*batch, N, c],
    shifts [Q + 1, *batch]); the result is (solves * weights).sum(0)."""
    import numpy as np
    from scipy.special import ellipj, ellipk

    if num_contour_quadrature is None:
        num_contour_quadrature = settings.num_contour_quadrature.value()
    output_batch_shape = torch.broadcast_shapes(linear_op.batch_shape, rhs.shape[:-2])
    preconditioner, preconditioner_lt, _ = linear_op._preconditioner()

    if preconditioner_lt is not None:  # sqrt_precond_matmul (:44-49): P^{1/2} rhs, itself by quadrature
        p_solves, p_weights, _, _ = contour_integral_quad(preconditioner_lt, rhs, inverse=False)
        rhs = (p_solves * p_weights).sum(0)

    if shifts is None:
        num_extra_dims = max(0, rhs.dim() - linear_op.dim())  # :56-64
        lanczos_init = rhs[(*([0] * num_extra_dims), Ellipsis, slice(None), slice(None, 1))]
        lanczos_init = lanczos_init.expand(*linear_op.shape[:-1], 1)
        with warnings.catch_warnings(), torch.no_grad():
            warnings.simplefilter("ignore", NumericalWarning)  # the CG run is stopped early on purpose
            _, lanczos_mat = linear_cg(
                linear_op._matmul, rhs=lanczos_init.contiguous(), n_tridiag=1, max_iter=max_lanczos_iter,
                tolerance=1e-5, max_tridiag_iter=max_lanczos_iter, preconditioner=preconditioner,
            )
            lanczos_mat = lanczos_mat.squeeze(0)
        try:  # approximate condition number from the Lanczos matrix (:85-96)
            approx_eigs = torch.linalg.eigvalsh(lanczos_mat)
            if approx_eigs.min() <= 0:
                raise RuntimeError
        except RuntimeError:
            approx_eigs = linear_op._diagonal()
        max_eig = approx_eigs.max(dim=-1)[0]
        min_eig = approx_eigs.min(dim=-1)[0]
        k2 = min_eig / max_eig

        # nodes and weights for every batch member at once (the reference loops over the members, :103-126; the
        # scipy functions are ufuncs, so the same double-precision values come out of one vectorised call)
        n_q = num_contour_quadrature
        k2_h = k2.flatten().double().cpu().numpy()
        min_h = min_eig.flatten().double().cpu().numpy()
        Kp = ellipk(1 - k2_h)  # complete elliptic integral of the first kind (:106)
        u = (np.arange(1, n_q + 1) - 0.5)[:, None] * Kp[None, :] / n_q  # imag(t), t = 1j (j - 1/2) K' / N
        sn, cn, dn, _ = ellipj(u, (1 - k2_h)[None, :])  # Jacobi elliptic functions (:109)
        cn = 1.0 / cn
        dn = dn * cn
        sn = 1j * sn * cn
        w = np.sqrt(min_h)[None, :] * sn
        w_pow2 = np.real(np.power(w, 2))
        constant = -2 * Kp * np.sqrt(min_h) / (math.pi * n_q)  # :118
        flat_shifts = torch.zeros(n_q + 1, k2.numel(), dtype=k2.dtype, device=k2.device)
        flat_shifts[1:] = torch.as_tensor(w_pow2, dtype=rhs.dtype).to(rhs.device)
        flat_weights = (torch.as_tensor(cn * dn, dtype=rhs.dtype) * torch.as_tensor(constant, dtype=rhs.dtype)).to(rhs.device)
        weights = flat_weights.view(n_q, *k2.shape, 1, 1)
        shifts = flat_shifts.view(n_q + 1, *k2.shape)
        shifts.sub_(shift_offset)
        if k2.shape != output_batch_shape:  # :132-134
            weights = torch.stack([w.expand(*output_batch_shape, 1, 1) for w in weights], 0)
            shifts = torch.stack([s.expand(output_batch_shape) for s in shifts], 0)

    with torch.no_grad():  # all shifted solves in one run (:137-144)
        solves = minres(linear_op._matmul, rhs, value=-1, shifts=shifts, preconditioner=preconditioner)
    no_shift_solves = solves[0]
    solves = solves[1:]
    if not inverse:  # one more product: K (K + t^2)^-1 b  (:147-148); the shifts ride in the column dimension
        q, c = solves.shape[0], solves.shape[-1]
        folded = solves.movedim(0, -2).reshape(*solves.shape[1:-1], q * c)  # [*batch, N, Q c]
        prod = linear_op._matmul(folded.contiguous())
        solves = prod.reshape(*prod.shape[:-1], q, c).movedim(-2, 0)
    return solves, weights, no_shift_solves, shifts


__all__ = ["contour_integral_quad"]
