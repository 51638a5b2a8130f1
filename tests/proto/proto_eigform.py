#!/usr/bin/env python3
"""Numerics prototype (numpy): the R-space CG of proto_rspace.py in the basis that diagonalises the preconditioned member.

With C^ = D^-1/2 C = U S V^T (E = C^T D^-1 C = V S^2 V^T), A^ = I + C^ C^^T and P^-1^ = I - C^ F C^^T both map span(U) to
itself and are the identity on its complement.  In U-coordinates A_U = I + S^2 =: Gam^2, P_U^-1 = I - S V^T F V S, and the
symmetric matrix Hs = Gam P_U^-1 Gam = Q Lam Q^T gives W = Gam^-1 Q Lam^1/2 with W^T P_U W = I, W^T A_U W = Lam.  In the
coordinates c of r^ = P^ [U W c + c' b_perp] (b_perp = the part of the right-hand side outside span(U), coordinate c'
starting at 1) linear_cg (linear_cg.py:245-332) is the CG of a DIAGONAL matrix:
    r.z = |c|^2 + c'^2 tau2      p.Ap = sum lam q^2 + q'^2 tau2      c -= alpha lam q     q = c + beta q     eta += alpha q
with c0 = Tin^T w0 (Tin = V S^-1 W), g0 = Tu^T w0 (Tu = V S^-1 W^-T), tau2 = s - c0.g0, and at the end
x = D^-1 (xi r0 + C y), xi = eta', y = Tin (eta - xi g0).  The residual norm (stop rule, records) is the one quantity that
is not diagonal; this prototype forms it as
    r^T r = del^T Nn del + 2 c' del.m0 + c'^2 a0,   del = c - c' c0,   Nn = W^-1 (U^T D U) W^-T,   m0 = Tu^T u0
(off the dependent chain: only the has_converged mask at 1e-10 feeds back); the kernel forms it in the coordinates of C
(g = Tu del, r^T r = c'^2 a0 + 2 c' g.u0 + g.G2 g): Nn amplifies the rounding of G2 by cond(E) (tools/fuzz_eigform.py).
Measured here: alphas / betas / residual norms / solutions against proto_rspace.cg_rspace and the exact solution on the
same cases, including rank-deficient C, r0 in span(C), small diagonals.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from proto_root_form import cases, f32, orc  # noqa: E402
from proto_w_recurrence import _root_form64  # noqa: E402
from proto_rspace import cg_rspace, exact  # noqa: E402

RANK_TOL = 1e-13
USE_EP = False


def eigform(E, F, G2):
    """Per-operator matrices of the diagonal form (fp64): Tin, Epinv, Nn, Tu [B,R,R], lam [B,R] (dropped directions: Tin col 0)."""
    B, R, _ = E.shape
    Tin = np.zeros((B, R, R)); Ep = np.zeros((B, R, R)); Nn = np.zeros((B, R, R)); Tu = np.zeros((B, R, R)); lam = np.ones((B, R))
    for b in range(B):
        s2, V = np.linalg.eigh(E[b])
        keep = s2 > RANK_TOL * s2.max()
        s2k, Vk = s2[keep], V[:, keep]
        r = int(keep.sum())
        S = np.sqrt(s2k)
        Gam = np.sqrt(1.0 + s2k)
        Pinv = np.eye(r) - (S[:, None] * (Vk.T @ F[b] @ Vk)) * S[None, :]
        Hs = Gam[:, None] * Pinv * Gam[None, :]
        Hs = (Hs + Hs.T) / 2
        lm, Q = np.linalg.eigh(Hs)
        W = (Q / Gam[:, None]) * np.sqrt(lm)[None, :]
        Winv_T = (Q * Gam[:, None]) / np.sqrt(lm)[None, :]          # W^-T = Gam Q Lam^-1/2
        VS = Vk / S[None, :]
        Tin[b, :, :r] = VS @ W
        Tu[b, :, :r] = VS @ Winv_T
        Ep[b] = (Vk / s2k[None, :]) @ Vk.T
        UDU = VS.T @ G2[b] @ VS
        Nn[b, :r, :r] = Winv_T.T @ UDU @ Winv_T
        lam[b, :r] = lm
    return Tin, Ep, Nn, Tu, lam


def cg_eig(C, dinv32, form, rhs, iters, eps=1e-10, stop_after=1e-10):
    Tin, Ep, Nn, Tu, lam = form
    C64 = C.astype(np.float64)
    di = dinv32.astype(np.float64)[..., None]
    b = rhs.astype(np.float64)
    Ct = np.swapaxes(C64, -1, -2)
    a0 = np.sum(b * b, -2, keepdims=True)
    nrm = np.sqrt(a0.astype(f32)).astype(f32)
    zero = nrm < eps
    nrm = np.where(zero, f32(1), nrm).astype(np.float64)
    w0 = Ct @ (b * di) / nrm
    u0 = Ct @ b / nrm
    s = np.sum(b * b * di, -2, keepdims=True) / nrm ** 2
    a0 = a0 / nrm ** 2
    dot = lambda a, bb: np.sum(a * bb, -2, keepdims=True)  # noqa: E731
    c0 = np.swapaxes(Tin, -1, -2) @ w0
    m0 = np.swapaxes(Tu, -1, -2) @ u0
    if USE_EP:   # first version: E^+ = V S^-2 V^T squares the conditioning of the basis (1e-4 errors at cond(E) = 1e12)
        e0 = Ep @ w0
        tau2 = np.maximum(s - dot(w0, e0), 0.0)
    else:        # g0 = Tu^T w0 = W^-1 beta0:  |beta0|^2 = c0.g0,  V S^-1 beta0 = Tin g0
        g0 = np.swapaxes(Tu, -1, -2) @ w0
        tau2 = np.maximum(s - dot(c0, g0), 0.0)
    lm = lam[..., None]
    c = c0.copy(); cp = np.ones_like(s)
    q = np.zeros_like(c); qp = np.zeros_like(s)
    eta = np.zeros_like(c); etap = np.zeros_like(s)
    rz = None
    alphas, betas, rns = [], [], []
    alpha = None
    for k in range(iters + 1):
        rzn = dot(c, c) + cp * cp * tau2
        dl = c - cp * c0
        s1 = dot(dl, Nn @ dl) + 2 * cp * dot(dl, m0) + cp * cp * a0
        rn = np.sqrt(np.maximum(s1, 0)).astype(f32)
        if rz is not None:
            beta = np.where(rz < eps, 0.0, rzn / np.where(rz < eps, 1.0, rz)).astype(f32).astype(np.float64)
            rn = np.where(zero, f32(0), rn)
            alphas.append(alpha[..., 0, :].astype(f32)); betas.append(beta[..., 0, :].astype(f32)); rns.append(rn[..., 0, :])
            if k == iters:
                break
        else:
            beta = np.zeros_like(s)
        conv = rn < stop_after
        rz = rzn
        q = c + beta * q
        qp = cp + beta * qp
        pAp = dot(lm * q, q) + qp * qp * tau2
        alpha = np.where(pAp < eps, 0.0, rz / np.where(pAp < eps, 1.0, pAp))
        alpha = np.where(conv, 0.0, alpha).astype(f32).astype(np.float64)
        eta = eta + alpha * q
        etap = etap + alpha * qp
        c = c - alpha * lm * q
        cp = cp - alpha * qp
    xi = etap
    y = (Tin @ eta - xi * e0) if USE_EP else Tin @ (eta - xi * g0)
    x = di * (xi * b + nrm * (C64 @ y))
    return x.astype(f32), np.stack(alphas), np.stack(betas), np.stack(rns)


def run(B, N, R, c, k, dscale, doff, seed=5, cscale=1.0, decay=0.0, inspan=False, dup=False, zerocol=False, tiny=0.0):
    C, d, rhs = cases.lowrank_diag(seed, B, N, R, c)
    C = (C * cscale).astype(f32)
    if decay:
        C = (C * (decay ** np.arange(R))[None, None, :]).astype(f32)
    if dup:      # exactly rank-deficient root: duplicated columns
        C[..., R // 2:] = C[..., :R - R // 2]
    if zerocol:
        C[..., 3] = 0
    if tiny:
        C[..., 5] *= f32(tiny)
    d = ((d - 0.5) * dscale + doff).astype(f32)
    if inspan:
        rng0 = np.random.default_rng(seed + 1)
        rhs = (C.astype(np.float64) @ rng0.standard_normal((B, R, c)) + 1e-3 * rhs).astype(f32)
    L, perm = orc.pivoted_cholesky(orc.LowRankRowSource(C), k)
    pre = orc.Preconditioner(L, d)
    x32, t32, info = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, n_tridiag=c, tolerance=1e-4,
                                   preconditioner=pre.apply)
    it = info.iterations
    C64 = C.astype(np.float64)
    dinv32 = (f32(1) / d).astype(f32)
    dq = 1.0 / dinv32.astype(np.float64)
    F64, E64 = _root_form64(C64, dq, L.astype(np.float64), perm)
    G2 = np.swapaxes(C64, -1, -2) @ C64
    rel = lambda a, bb: float(np.max(np.linalg.norm(a - bb, axis=-2) / np.linalg.norm(bb, axis=-2)))  # noqa: E731
    rmax = lambda a, bb: float(np.max(np.abs(a - bb) / np.maximum(np.abs(bb), 1e-30)))  # noqa: E731
    xe = exact(C, d, rhs)
    xr, al, be, rn = cg_rspace(C, dinv32, F64, E64, G2, rhs, it)
    form = eigform(E64, F64, G2)
    xg, al2, be2, rn2 = cg_eig(C, dinv32, form, rhs, it)
    print(f"N={N} R={R} k={k} Cx{cscale} decay={decay} d in [{doff:g},{doff + dscale:g}] inspan={inspan} dup={dup} zerocol={zerocol} tiny={tiny}: iters {it}\n"
          f"   x vs exact: oracle32 {rel(x32, xe):.1e} rspace {rel(xr, xe):.1e} eig {rel(xg, xe):.1e} | eig vs rspace {rel(xg, xr):.1e}"
          f" | alpha {rmax(al2, al):.1e} beta {rmax(be2, be):.1e} rn {rmax(rn2, rn):.1e} (last rn {rn[-1].max():.1e} / {rn2[-1].max():.1e})")


if __name__ == "__main__":
    run(4, 8192, 32, 3, 15, 1.0, 0.5)
    run(4, 8192, 32, 3, 15, 0.1, 0.01)
    run(4, 2048, 32, 3, 15, 0.01, 0.001)
    run(4, 4096, 16, 3, 7, 1.0, 0.05)
    run(3, 4096, 8, 3, 15, 1.0, 0.5)
    run(3, 8192, 32, 3, 15, 1.0, 0.5, cscale=10.0)
    run(3, 8192, 32, 3, 4, 1.0, 0.5)
    run(3, 4096, 32, 3, 15, 1.0, 0.5, decay=0.6)
    run(3, 4096, 32, 3, 15, 0.01, 0.001, decay=0.7)
    run(3, 4096, 32, 3, 15, 1.0, 0.5, inspan=True)
    run(3, 4096, 32, 3, 15, 0.01, 0.001, inspan=True)
    run(3, 4096, 32, 3, 15, 1.0, 0.5, dup=True)
    run(3, 4096, 32, 3, 15, 1.0, 0.5, zerocol=True)
    run(3, 4096, 32, 3, 15, 0.01, 0.001, dup=True, inspan=True)
    for use_ep in (True, False):   # one column of C scaled by 1e-6: cond(E) = 1e12, the direction is kept
        globals()["USE_EP"] = use_ep
        print("USE_EP", use_ep)
        run(3, 8192, 12, 3, 12, 0.01, 0.001, tiny=1e-6)
        run(3, 4096, 32, 3, 15, 0.01, 0.001, tiny=1e-6)
        run(3, 4096, 32, 3, 15, 1.0, 0.5, tiny=1e-5)
