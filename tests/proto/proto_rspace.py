#!/usr/bin/env python3
"""Numerics prototype (numpy): the WHOLE preconditioned CG of A = C C^T + D, P = L L^T + D (L = C M) in R-space.

With G = I - F - E F the iteration matrix is A P^-1 = I + C G C^T D^-1, so every vector CG ever forms is a combination of
the right-hand side and the columns of C:

    r_k = rho r0 + C g,      p_k = D^-1 (pi r0 + C h),      x_k = D^-1 (xi r0 + C y)            (rho, pi, xi scalars; g, h, y in R^R)

and every inner product follows from ONE reduction over the rows per solve,
    w0 = C^T D^-1 r0,   u0 = C^T r0,   s = r0^T D^-1 r0,   a0 = r0^T r0
and two Gram matrices of the operator, E = C^T D^-1 C and G2 = C^T C (cached with the preconditioner, fp64):
    w = C^T D^-1 r = rho w0 + E g            r^T D^-1 r = rho^2 s + rho g.w0 + g.w            r^T r = rho^2 a0 + 2 rho g.u0 + g.G2 g
    z = P^-1 r = D^-1 (rho r0 + C (g - F w))          r.z = r^T D^-1 r - w.F w
    t = C^T p = pi w0 + E h                           A p = pi r0 + C (h + t)
    p^T A p = |t|^2 + (pi^2 s + 2 pi h.w0 + h.E h)
The 11 iterations need no row of C at all; the rows come back once, for x = D^-1 (xi r0 + C y) (in fp64: for small
diagonals the two terms cancel, which is what broke the two-product variant of proto_root_form.py in fp32).
This file measures: solution against the fp64 iteration / the oracle's fp32 iteration / the exact solution, and how
accurate the Gram matrices have to be (fp64-exact, 1e-12 noise, fp32-level noise).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from proto_root_form import cases, f32, orc  # noqa: E402
from proto_w_recurrence import _root_form64  # noqa: E402


def cg_rspace(C, dinv32, F, E, G2, rhs, iters, eps=1e-10, stop_after=1e-10, coef32=True):
    """C fp32 [B,N,R], dinv32 fp32 [B,N], F / E / G2 fp64 [B,R,R], rhs fp32 [B,N,c].  Returns x (fp32), alphas, betas, rns."""
    C64 = C.astype(np.float64)
    di = dinv32.astype(np.float64)[..., None]
    b = rhs.astype(np.float64)
    Ct = np.swapaxes(C64, -1, -2)
    # ---- the one reduction over the rows (on the RAW column; scaled by 1 / norm afterwards) ----
    a0 = np.sum(b * b, -2, keepdims=True)
    nrm = np.sqrt(a0.astype(f32)).astype(f32)          # rhs.norm: fp32 like the reference (:177)
    zero = nrm < eps
    nrm = np.where(zero, f32(1), nrm).astype(np.float64)
    w0 = Ct @ (b * di) / nrm
    u0 = Ct @ b / nrm
    s = np.sum(b * b * di, -2, keepdims=True) / nrm ** 2
    a0 = a0 / nrm ** 2
    R, c = C.shape[-1], rhs.shape[-1]
    shp = C.shape[:-2] + (R, c)
    rho = np.ones_like(s); g = np.zeros(shp)
    pi = np.zeros_like(s); h = np.zeros(shp); Eh = np.zeros(shp)
    xi = np.zeros_like(s); y = np.zeros(shp)
    dot = lambda a, bb: np.sum(a * bb, -2, keepdims=True)  # noqa: E731
    rz = None
    beta = np.zeros_like(s)
    tt_old = np.zeros_like(s); dpp = np.zeros_like(s); t_old = np.zeros(shp)
    alphas, betas, rns = [], [], []
    conv = None
    for k in range(iters + 1):
        Eg = E @ g
        w = rho * w0 + Eg
        v = F @ w
        Ev = E @ v
        s2 = rho * rho * s + rho * dot(g, w0) + dot(g, w)
        s1 = rho * rho * a0 + 2 * rho * dot(g, u0) + dot(g, G2 @ g)
        rzn = s2 - dot(w, v)
        rn = np.sqrt(np.maximum(s1, 0)).astype(f32)
        if rz is not None:
            beta = np.where(rz < eps, 0.0, rzn / np.where(rz < eps, 1.0, rz))
            if coef32:
                beta = beta.astype(f32).astype(np.float64)
            rn = np.where(zero, f32(0), rn)
            alphas.append(alpha[..., 0, :].astype(f32)); betas.append(beta[..., 0, :].astype(f32)); rns.append(rn[..., 0, :])
            if k == iters:
                break
        conv = rn < stop_after
        rz = rzn
        zc = g - v                                   # z = D^-1 (rho r0 + C zc)
        tz = w - Ev                                  # C^T z
        # sum d z^2 = s2 - 2 w.v + v.Ev ; sum d z p_old = r^T p_old - v.t_old,  r^T p_old = rho pi s + rho h.w0 + pi g.w0 + g.Eh
        dzz = s2 - 2 * dot(w, v) + dot(v, Ev)
        rp = rho * pi * s + rho * dot(h, w0) + pi * dot(g, w0) + dot(g, Eh)
        dzp = rp - dot(v, t_old)
        dpp = dzz + 2 * beta * dzp + beta * beta * dpp
        tt_old = dot(tz, tz) + 2 * beta * dot(tz, t_old) + beta * beta * tt_old
        t_old = tz + beta * t_old
        pi = rho + beta * pi
        h = zc + beta * h
        Eh = (Eg - Ev) + beta * Eh
        pAp = tt_old + dpp
        alpha = np.where(pAp < eps, 0.0, rz / np.where(pAp < eps, 1.0, pAp))
        alpha = np.where(conv, 0.0, alpha)
        if coef32:
            alpha = alpha.astype(f32).astype(np.float64)
        xi = xi + alpha * pi
        y = y + alpha * h
        rho = rho - alpha * pi
        g = g - alpha * (h + t_old)
    x = di * (xi * b + nrm * (C64 @ y))              # = nrm * D^-1 (xi r0 + C y)
    return x.astype(f32), np.stack(alphas), np.stack(betas), np.stack(rns)


def noisy(Mx, rel, rng):
    n = rng.standard_normal(Mx.shape)
    n = (n + np.swapaxes(n, -1, -2)) / 2
    return Mx * (1 + rel * n)


def exact(C, d, rhs):
    C64, d64, r64 = C.astype(np.float64), d.astype(np.float64)[..., None], rhs.astype(np.float64)
    Cd = C64 / d64
    cap = np.eye(C.shape[-1]) + np.swapaxes(C64, -1, -2) @ Cd
    return r64 / d64 - Cd @ np.linalg.solve(cap, np.swapaxes(C64, -1, -2) @ (r64 / d64))


def run(B, N, R, c, k, dscale, doff, seed=5, cscale=1.0, decay=0.0, inspan=False):
    C, d, rhs = cases.lowrank_diag(seed, B, N, R, c)
    C = (C * cscale).astype(f32)
    if decay:  # columns of C with geometrically decaying norms: an ill-conditioned Gram matrix E
        C = (C * (decay ** np.arange(R))[None, None, :]).astype(f32)
    d = ((d - 0.5) * dscale + doff).astype(f32)
    if inspan:  # right-hand sides (almost) inside the column space of C: the Gram matrix of [r0, C] is near-singular
        rng0 = np.random.default_rng(seed + 1)
        rhs = (C.astype(np.float64) @ rng0.standard_normal((B, R, c)) + 1e-3 * rhs).astype(f32)
    L, perm = orc.pivoted_cholesky(orc.LowRankRowSource(C), k)
    pre = orc.Preconditioner(L, d)
    x32, t32, info = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, n_tridiag=c, tolerance=1e-4,
                                   preconditioner=pre.apply)
    C64, d64 = C.astype(np.float64), d.astype(np.float64)
    pre64 = orc.Preconditioner(L.astype(np.float64), d64)
    it = info.iterations
    x64, t64, info64 = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C64, d64, v), rhs.astype(np.float64), n_tridiag=c,
                                     tolerance=0.0, max_iter=it, preconditioner=pre64.apply)
    dinv32 = (f32(1) / d).astype(f32)
    dq = 1.0 / dinv32.astype(np.float64)             # the diagonal the R-space iteration actually uses
    F64, E64 = _root_form64(C64, dq, L.astype(np.float64), perm)
    G2 = np.swapaxes(C64, -1, -2) @ C64
    rel = lambda a, bb: float(np.max(np.linalg.norm(a - bb, axis=-2) / np.linalg.norm(bb, axis=-2)))  # noqa: E731
    xe = exact(C, d, rhs)
    xr, al, be, rn = cg_rspace(C, dinv32, F64, E64, G2, rhs, it)
    rng = np.random.default_rng(1)
    out = [f"N={N} R={R} k={k} Cx{cscale} decay={decay} d in [{doff:g},{doff + dscale:g}] inspan={inspan}: iters {it} (fp64 run {info64.iterations})",
           f"   x vs exact: oracle32 {rel(x32, xe):.1e} oracle64 {rel(x64, xe):.1e} rspace {rel(xr, xe):.1e} | rspace vs oracle32 {rel(xr, x32):.1e} vs oracle64 {rel(xr, x64):.1e}"
           f" | final rn rspace {rn[-1].max():.1e}"]
    for lvl in (1e-12, 1e-9, 1e-7):
        xn, _, _, rnn = cg_rspace(C, dinv32, F64, noisy(E64, lvl, rng), noisy(G2, lvl, rng), rhs, it)
        out.append(f"   Gram noise {lvl:g}: x vs exact {rel(xn, xe):.1e} vs oracle32 {rel(xn, x32):.1e} final rn {rnn[-1].max():.1e}")
    print("\n".join(out))


if __name__ == "__main__":
    run(4, 8192, 32, 3, 15, 1.0, 0.5)            # the headline spectrum
    run(4, 8192, 32, 3, 15, 0.1, 0.01)
    run(4, 2048, 32, 3, 15, 0.01, 0.001)         # small diagonals
    run(4, 4096, 16, 3, 7, 1.0, 0.05)
    run(3, 4096, 8, 3, 15, 1.0, 0.5)             # rank <= pivots: P = A
    run(3, 8192, 32, 3, 15, 1.0, 0.5, cscale=10.0)   # strong low-rank part
    run(3, 8192, 32, 3, 4, 1.0, 0.5)             # weak preconditioner
    run(3, 4096, 32, 3, 15, 1.0, 0.5, decay=0.6)     # E with condition number ~ 1e13
    run(3, 4096, 32, 3, 15, 0.01, 0.001, decay=0.7)
    run(3, 4096, 32, 3, 15, 1.0, 0.5, inspan=True)   # r0 almost inside span(C)
    run(3, 4096, 32, 3, 15, 0.01, 0.001, inspan=True)
