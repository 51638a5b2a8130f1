#!/usr/bin/env python3
"""Numerics prototype (numpy) of the ROOT-FORM preconditioner the operator-resident kernels use for
A = C C^T + D with the pivoted-Cholesky preconditioner P = L L^T + D.

Every column of the pivoted-Cholesky factor of K = C C^T lies in the column space of C:  L = C M with the R x m matrix
    M[:, j] = (C[pi_j, :]^T - sum_{i<j} M[:, i] L[pi_j, i]) / L[pi_j, j]       (_pivoted_cholesky.py:77-92 written for M)
so with E = C^T D^-1 C (R x R) and F = M (I + M^T E M)^-1 M^T (R x R, symmetric)
    P^-1 r = D^-1 (r - C F w),   w = C^T D^-1 r,          logdet P = logdet(I + M^T E M) + sum log d
and the quantities of the single-reduction iteration (tools/proto_single_reduction.py) need NO second tall matrix:
    r.z = s2 - w.v,   v = F w,    C^T z = w - E v,    sum d z^2 = s2 - 2 w.v + v.E v,    sum d z p_old = rp - v.(C^T p_old)
One reduction of {w (R values), s1, s2, rp} per iteration; the per-row work is three passes over the rows of C.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
from oracle import lo_oracle as orc  # noqa: E402

f32 = np.float32


def root_form(C, d, L, perm):
    """F, E (fp32) and logdet P from the factor's pivot rows, in fp64."""
    C64, d64, L64 = C.astype(np.float64), d.astype(np.float64), L.astype(np.float64)
    B, N, R = C.shape
    m = L.shape[-1]
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
    F = M @ np.linalg.solve(G, np.swapaxes(M, -1, -2))
    logdet = np.linalg.slogdet(G)[1] + np.log(d64).sum(-1)
    return F.astype(f32), E.astype(f32), logdet, M


def cg_root_form(C, d, F, E, rhs, iters, eps=1e-10, stop_after=1e-10):
    dt = C.dtype
    Ct = np.swapaxes(C, -1, -2)
    dcol = d[..., None]
    dinv = (1 / dcol).astype(dt)
    nrm = np.sqrt(np.sum(rhs * rhs, axis=-2, keepdims=True, dtype=dt))
    r = rhs / nrm
    x = np.zeros_like(r)
    p = np.zeros_like(r)
    t = np.zeros(C.shape[:-2] + (C.shape[-1], rhs.shape[-1]), dt)
    dpp = np.zeros_like(nrm)
    beta = np.zeros_like(nrm)
    rz = None

    def reduce_all(r, p):
        rd = r * dinv
        return Ct @ rd, np.sum(r * r, -2, keepdims=True, dtype=dt), np.sum(r * rd, -2, keepdims=True, dtype=dt), \
            np.sum(r * p, -2, keepdims=True, dtype=dt)

    w, s1, s2, rp = reduce_all(r, p)
    conv = np.sqrt(s1) < stop_after
    alphas, betas, rns = [], [], []
    for k in range(iters + 1):
        v = F @ w
        Ev = E @ v
        wv = np.sum(w * v, -2, keepdims=True, dtype=dt)
        rzn = s2 - wv
        if rz is not None:
            beta = np.where(rz < eps, f32(0), rzn / np.where(rz < eps, f32(1), rz)).astype(dt)
            rn = np.sqrt(s1)
            conv = rn < stop_after
            alphas.append(alpha[..., 0, :]); betas.append(beta[..., 0, :]); rns.append(rn[..., 0, :])
            if k == iters:
                break
        rz = rzn
        dzz = s2 - 2 * wv + np.sum(v * Ev, -2, keepdims=True, dtype=dt)
        dzp = rp - np.sum(v * t, -2, keepdims=True, dtype=dt)
        dpp = dzz + 2 * beta * dzp + beta * beta * dpp
        t = (w - Ev) + beta * t
        p = beta * p + (r - C @ v) * dinv
        pAp = np.sum(t * t, -2, keepdims=True, dtype=dt) + dpp
        alpha = np.where(pAp < eps, f32(0), rz / np.where(pAp < eps, f32(1), pAp))
        alpha = np.where(conv, f32(0), alpha).astype(dt)
        x = x + alpha * p
        r = r - alpha * (C @ t + dcol * p)
        w, s1, s2, rp = reduce_all(r, p)
    return x * nrm, np.stack(alphas), np.stack(betas), np.stack(rns)


def cg_two_product(C, d, F, E, rhs, iters, eps=1e-10, stop_after=1e-10):
    """REJECTED variant: TWO products with C per iteration instead of three -- A p by the recurrence
    A p_new = C (w - E v - v) + r + beta A p_old, p and x kept as (vector - C small)/d pairs and x formed at the end.
    A third fewer FMAs, but for small diagonals the final x = (xr - C xv)/d cancels: 2e-4 relative error at
    d ~ 1e-3 against 7e-6 for the three-product form (printed by run())."""
    dt = C.dtype
    Ct = np.swapaxes(C, -1, -2)
    dcol = d[..., None]
    dinv = (1 / dcol).astype(dt)
    nrm = np.sqrt(np.sum(rhs * rhs, axis=-2, keepdims=True, dtype=dt))
    r = rhs / nrm
    xr = np.zeros_like(r); pr = np.zeros_like(r); q = np.zeros_like(r)
    Rk = C.shape[-1]; c = rhs.shape[-1]
    t = np.zeros(C.shape[:-2] + (Rk, c), dt); pv = np.zeros_like(t); xv = np.zeros_like(t)
    dpp = np.zeros_like(nrm); beta = np.zeros_like(nrm); rz = None
    def reduce_all(r, pr):
        rd = r * dinv
        return Ct @ rd, np.sum(r * r, -2, keepdims=True, dtype=dt), np.sum(r * rd, -2, keepdims=True, dtype=dt), \
            np.sum(rd * pr, -2, keepdims=True, dtype=dt)
    w, s1, s2, rpr = reduce_all(r, pr)
    conv = np.sqrt(s1) < stop_after
    for k in range(iters + 1):
        v = F @ w
        Ev = E @ v
        wv = np.sum(w * v, -2, keepdims=True, dtype=dt)
        rzn = s2 - wv
        if rz is not None:
            beta = np.where(rz < eps, f32(0), rzn / np.where(rz < eps, f32(1), rz)).astype(dt)
            conv = np.sqrt(s1) < stop_after
            if k == iters:
                break
        rz = rzn
        rp = rpr - np.sum(w * pv, -2, keepdims=True, dtype=dt)
        dzz = s2 - 2 * wv + np.sum(v * Ev, -2, keepdims=True, dtype=dt)
        dzp = rp - np.sum(v * t, -2, keepdims=True, dtype=dt)
        dpp = dzz + 2 * beta * dzp + beta * beta * dpp
        zc = w - Ev
        t = zc + beta * t
        pr = r + beta * pr
        pv = v + beta * pv
        pAp = np.sum(t * t, -2, keepdims=True, dtype=dt) + dpp
        alpha = np.where(pAp < eps, f32(0), rz / np.where(pAp < eps, f32(1), pAp))
        alpha = np.where(conv, f32(0), alpha).astype(dt)
        xr = xr + alpha * pr
        xv = xv + alpha * pv
        q = C @ (zc - v) + r + beta * q
        r = r - alpha * q
        w, s1, s2, rpr = reduce_all(r, pr)
    x = (xr - C @ xv) * dinv
    return x * nrm


def run(B, N, R, c, k, dscale, doff, seed=5):
    C, d, rhs = cases.lowrank_diag(seed, B, N, R, c)
    d = ((d - 0.5) * dscale + doff).astype(f32)
    L, perm = orc.pivoted_cholesky(orc.LowRankRowSource(C), k)
    pre = orc.Preconditioner(L, d)
    F, E, logdet, M = root_form(C, d, L, perm)
    mm = lambda v: orc.matvec_lowrank_diag(C, d, v)  # noqa: E731
    x32, t32, info = orc.linear_cg(mm, rhs, n_tridiag=min(c, 16), tolerance=1e-4, preconditioner=pre.apply)
    C64, d64 = C.astype(np.float64), d.astype(np.float64)
    pre64 = orc.Preconditioner(L.astype(np.float64), d64)
    x64, _, _ = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C64, d64, v), rhs.astype(np.float64),
                              n_tridiag=min(c, 16), tolerance=1e-4, preconditioner=pre64.apply)
    xs, al, be, rn = cg_root_form(C, d, F, E, rhs, info.iterations)
    x2 = cg_two_product(C, d, F, E, rhs, info.iterations)
    rel = lambda a, b: float(np.max(np.linalg.norm(a - b, axis=-2) / np.linalg.norm(b, axis=-2)))  # noqa: E731
    # preconditioner itself: apply both forms to the rhs
    z_q = pre.apply(rhs)
    w = np.swapaxes(C, -1, -2) @ (rhs / d[..., None])
    z_f = (rhs - C @ (F @ w)) / d[..., None]
    print(f"N={N} R={R} k={k} d in [{doff},{doff + dscale}]: iters {info.iterations} | x: orc32-64 {rel(x32, x64):.2e} "
          f"root32-64 {rel(xs, x64):.2e} (two-product variant {rel(x2, x64):.2e}) root-orc {rel(xs, x32):.2e} | P^-1 rhs Q-form vs root-form {rel(z_f, z_q):.2e} | "
          f"logdet_p {np.abs(logdet - pre.logdet).max():.2e} of {np.abs(pre.logdet).max():.1f} | L - C M {np.abs(C.astype(np.float64) @ M - L).max():.2e}")


if __name__ == "__main__":
    run(4, 8192, 32, 3, 15, 1.0, 0.5)
    run(4, 8192, 32, 3, 15, 0.1, 0.01)
    run(4, 2048, 32, 3, 15, 0.01, 0.001)
    run(4, 4096, 16, 3, 7, 1.0, 0.05)
    run(2, 8192, 32, 3, 15, 10.0, 0.5)
    run(3, 4096, 8, 3, 15, 1.0, 0.5)
