#!/usr/bin/env python3
"""Numerics prototype (numpy): the root-form iteration of proto_root_form.py with w = C^T D^-1 r carried by RECURRENCE.

    r' = r - alpha (C t + d o p),  t = C^T p     =>     w' = C^T D^-1 r' = w - alpha (E t + t),   E = C^T D^-1 C

i.e. the third pass over the rows of C (the partials of w) and 32 of the 35 values of the per-iteration all-reduce
disappear; what is still reduced over the rows are the three scalars {sum r^2, sum r^2/d, sum r o p}.  Identical in
exact arithmetic; w then carries its own rounding history (like r itself does in every CG), which is what this file
measures against the fp64 iteration and the oracle's fp32 iteration: solutions, CG coefficients, residual norms.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from proto_root_form import cases, cg_root_form, f32, orc, root_form  # noqa: E402


def cg_w_recurrence(C, d, F, E, rhs, iters, eps=1e-10, stop_after=1e-10, refresh=0, small64=False):
    """small64: the R-space quantities (w, t, v, E v and the matrices E, F) in fp64, the N-vectors and the three row
    reductions in fp32."""
    dt = C.dtype
    sdt = np.float64 if small64 else dt
    Ct = np.swapaxes(C, -1, -2)
    dcol = d[..., None]
    dinv = (1 / dcol).astype(dt)
    nrm = np.sqrt(np.sum(rhs * rhs, axis=-2, keepdims=True, dtype=dt))
    r = rhs / nrm
    x = np.zeros_like(r)
    p = np.zeros_like(r)
    t = np.zeros(C.shape[:-2] + (C.shape[-1], rhs.shape[-1]), sdt)
    dpp = np.zeros_like(nrm)
    beta = np.zeros_like(nrm)
    rz = None
    F = F.astype(sdt)
    E = E.astype(sdt)

    def scalars(r, p):
        rd = r * dinv
        return np.sum(r * r, -2, keepdims=True, dtype=dt), np.sum(r * rd, -2, keepdims=True, dtype=dt), \
            np.sum(r * p, -2, keepdims=True, dtype=dt)

    w = (Ct.astype(sdt) @ (r * dinv).astype(sdt))  # the ONE full product for w (rides on the load of the member)
    s1, s2, rp = scalars(r, p)
    conv = np.sqrt(s1) < stop_after
    alphas, betas, rns = [], [], []
    for k in range(iters + 1):
        v = F @ w
        Ev = E @ v
        wv = np.sum(w * v, -2, keepdims=True, dtype=sdt)
        rzn = (s2 - wv).astype(dt)
        if rz is not None:
            beta = np.where(rz < eps, f32(0), rzn / np.where(rz < eps, f32(1), rz)).astype(dt)
            rn = np.sqrt(s1)
            conv = rn < stop_after
            alphas.append(alpha[..., 0, :]); betas.append(beta[..., 0, :]); rns.append(rn[..., 0, :])
            if k == iters:
                break
        rz = rzn
        dzz = (s2 - 2 * wv + np.sum(v * Ev, -2, keepdims=True, dtype=sdt)).astype(dt)
        dzp = (rp - np.sum(v * t, -2, keepdims=True, dtype=sdt)).astype(dt)
        dpp = dzz + 2 * beta * dzp + beta * beta * dpp
        t = (w - Ev) + beta * t
        p = beta * p + (r - C @ v.astype(dt)) * dinv
        pAp = (np.sum(t * t, -2, keepdims=True, dtype=sdt) + dpp).astype(dt)
        alpha = np.where(pAp < eps, f32(0), rz / np.where(pAp < eps, f32(1), pAp))
        alpha = np.where(conv, f32(0), alpha).astype(dt)
        x = x + alpha * p
        r = r - alpha * (C @ t.astype(dt) + dcol * p)
        if refresh and (k + 1) % refresh == 0:
            w = (Ct @ (r * dinv)).astype(sdt)
        else:
            w = w - alpha * (E @ t + t)
        s1, s2, rp = scalars(r, p)
    return x * nrm, np.stack(alphas), np.stack(betas), np.stack(rns)


def _root_form64(C64, d64, L64, perm):
    """F, E of root_form without the rounding to fp32."""
    B, N, R = C64.shape
    m = L64.shape[-1]
    M = np.zeros((B, R, m))
    for b in range(B):
        for j in range(m):
            pj = perm[b, j]
            col = C64[b, pj, :].copy()
            for i in range(j):
                col -= M[b, :, i] * L64[b, pj, i]
            M[b, :, j] = col / L64[b, pj, j]
    E = np.swapaxes(C64, -1, -2) @ (C64 / d64[..., None])
    G = np.eye(m) + np.swapaxes(M, -1, -2) @ E @ M
    return M @ np.linalg.solve(G, np.swapaxes(M, -1, -2)), E


def run(B, N, R, c, k, dscale, doff, seed=5, cscale=1.0):
    C, d, rhs = cases.lowrank_diag(seed, B, N, R, c)
    C = (C * cscale).astype(f32)
    d = ((d - 0.5) * dscale + doff).astype(f32)
    L, perm = orc.pivoted_cholesky(orc.LowRankRowSource(C), k)
    pre = orc.Preconditioner(L, d)
    F, E, logdet, M = root_form(C, d, L, perm)
    x32, t32, info = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, n_tridiag=c, tolerance=1e-4,
                                   preconditioner=pre.apply)
    C64, d64 = C.astype(np.float64), d.astype(np.float64)
    pre64 = orc.Preconditioner(L.astype(np.float64), d64)
    x64, t64, _ = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C64, d64, v), rhs.astype(np.float64), n_tridiag=c,
                                tolerance=1e-4, preconditioner=pre64.apply)
    it = info.iterations
    xa, al_a, be_a, rn_a = cg_root_form(C, d, F, E, rhs, it)
    xb, al_b, be_b, rn_b = cg_w_recurrence(C, d, F, E, rhs, it)
    F64, E64, _, _ = root_form(C64, d64, L.astype(np.float64), perm)
    F64d, E64d = _root_form64(C64, d64, L.astype(np.float64), perm)
    xc, al_c, be_c, rn_c = cg_w_recurrence(C, d, F64d, E64d, rhs, it, small64=True)
    x_ref, al_r, be_r, rn_r = cg_root_form(C64, d64, F64.astype(np.float64), E64.astype(np.float64), rhs.astype(np.float64), it)
    rel = lambda a, b: float(np.max(np.linalg.norm(a - b, axis=-2) / np.linalg.norm(b, axis=-2)))  # noqa: E731

    def coef_err(al, be, lead):  # alpha / beta of the first `lead` iterations against the fp64 iteration
        ea = np.abs(al[:lead] - al_r[:lead]) / np.maximum(np.abs(al_r[:lead]), 1e-30)
        eb = np.abs(be[:lead] - be_r[:lead]) / np.maximum(np.abs(be_r[:lead]), 1e-30)
        return float(ea.max()), float(eb.max())

    # iterations until the fp64 residual has dropped below 1e-5 (beyond: fp32 noise in every formulation)
    lead = int(np.argmax((rn_r.max(axis=(1, 2)) < 1e-5))) or it
    print(f"N={N} R={R} k={k} C x{cscale} d in [{doff:g},{doff + dscale:g}]: iters {it}, meaningful {lead} | x vs fp64: oracle "
          f"{rel(x32, x64):.1e} three-pass {rel(xa, x64):.1e} w-recurrence {rel(xb, x64):.1e} | alpha/beta vs fp64 (first {lead}): "
          f"three-pass {coef_err(al_a, be_a, lead)[0]:.1e}/{coef_err(al_a, be_a, lead)[1]:.1e} w-rec "
          f"{coef_err(al_b, be_b, lead)[0]:.1e}/{coef_err(al_b, be_b, lead)[1]:.1e} | final resid three-pass {rn_a[-1].max():.1e} "
          f"w-rec {rn_b[-1].max():.1e} fp64 {rn_r[-1].max():.1e}\n      w-recurrence with the R-space algebra in fp64: x {rel(xc, x64):.1e}, "
          f"alpha/beta {coef_err(al_c, be_c, lead)[0]:.1e}/{coef_err(al_c, be_c, lead)[1]:.1e}, final resid {rn_c[-1].max():.1e}")


if __name__ == "__main__":
    run(4, 8192, 32, 3, 15, 1.0, 0.5)            # the headline spectrum
    run(4, 8192, 32, 3, 15, 0.1, 0.01)
    run(4, 2048, 32, 3, 15, 0.01, 0.001)         # small diagonals (where the two-product variant failed)
    run(4, 4096, 16, 3, 7, 1.0, 0.05)
    run(2, 8192, 32, 3, 15, 10.0, 0.5)
    run(3, 4096, 8, 3, 15, 1.0, 0.5)             # rank <= pivots: P = A
    run(3, 8192, 32, 3, 15, 1.0, 0.5, cscale=0.05)   # well-conditioned (g23 recipe)
    run(3, 8192, 32, 3, 15, 1.0, 0.5, cscale=10.0)   # strong low-rank part
    run(3, 8192, 32, 3, 4, 1.0, 0.5)             # weak preconditioner: many meaningful iterations
    run(3, 8192, 32, 3, 1, 1.0, 0.5)
