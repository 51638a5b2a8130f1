#!/usr/bin/env python3
"""Numerics prototype (numpy, fp32) of the single-reduction form of the preconditioned CG iteration the
operator-resident kernels use for A = C C^T + diag(d) with the Woodbury preconditioner z = r/d - Q (Q^T r).

Per iteration ONE reduction over the rows delivers
    u = Q^T r,  w = C^T (r/d),  s1 = sum r^2,  s2 = sum r^2/d,  rp = sum r o p_old
and everything else the reference's iteration needs (linear_cg.py:245-300) follows from small per-member
recurrences with H = C^T Q and G = Q^T D Q formed once per member:
    r.z        = s2 - |u|^2
    C^T p_new  = (w - H u) + beta C^T p_old
    Q^T D p_new= (u - G u) + beta Q^T D p_old
    sum d p_new^2 = dzz + 2 beta dzp + beta^2 sum d p_old^2,  dzz = s2 - 2|u|^2 + u^T G u,  dzp = rp - u^T (Q^T D p_old)
    p.Ap       = |C^T p|^2 + sum d p^2
Compared here against the oracle (same arithmetic as the reference, fp32) and an fp64 run of it.
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


def single_reduction_cg(C, d, Q, rhs, iters, eps=1e-10, stop_after=1e-10):
    """All arrays fp32; C [B,N,R], d [B,N], Q [B,N,k], rhs [B,N,c].  Returns x, alphas, betas, resid norms."""
    dt = C.dtype
    Ct, Qt = np.swapaxes(C, -1, -2), np.swapaxes(Q, -1, -2)
    dcol = d[..., None]
    dinv = (1 / dcol).astype(dt)
    H = Ct @ Q  # [B,R,k]
    G = Qt @ (dcol * Q)  # [B,k,k]
    nrm = np.sqrt(np.sum(rhs * rhs, axis=-2, keepdims=True, dtype=dt))
    r = rhs / nrm
    x = np.zeros_like(r)

    def reduce_all(r, p):
        rd = r * dinv
        u = Qt @ r
        w = Ct @ rd
        s1 = np.sum(r * r, axis=-2, keepdims=True, dtype=dt)
        s2 = np.sum(r * rd, axis=-2, keepdims=True, dtype=dt)
        rp = np.sum(r * p, axis=-2, keepdims=True, dtype=dt) if p is not None else None
        return u, w, s1, s2, rp

    u, w, s1, s2, _ = reduce_all(r, None)
    uu = np.sum(u * u, axis=-2, keepdims=True, dtype=dt)
    rz = s2 - uu
    z = r * dinv - Q @ u
    p = z.copy()
    t = w - H @ u
    Gu = G @ u
    g = u - Gu
    dpp = s2 - 2 * uu + np.sum(u * Gu, axis=-2, keepdims=True, dtype=dt)
    conv = np.sqrt(s1) < stop_after
    alphas, betas, rns = [], [], []
    for k in range(iters):
        pAp = np.sum(t * t, axis=-2, keepdims=True, dtype=dt) + dpp
        alpha = np.where(pAp < eps, f32(0), rz / np.where(pAp < eps, f32(1), pAp))
        alpha = np.where(conv, f32(0), alpha).astype(dt)
        r = r - alpha * (C @ t + dcol * p)
        x = x + alpha * p
        u, w, s1, s2, rp = reduce_all(r, p)
        uu = np.sum(u * u, axis=-2, keepdims=True, dtype=dt)
        rzn = s2 - uu
        beta = np.where(rz < eps, f32(0), rzn / np.where(rz < eps, f32(1), rz)).astype(dt)
        rz = rzn
        z = r * dinv - Q @ u
        Gu = G @ u
        dzz = s2 - 2 * uu + np.sum(u * Gu, axis=-2, keepdims=True, dtype=dt)
        dzp = rp - np.sum(u * g, axis=-2, keepdims=True, dtype=dt)
        dpp = dzz + 2 * beta * dzp + beta * beta * dpp
        t = (w - H @ u) + beta * t
        g = (u - Gu) + beta * g
        p = z + beta * p
        rn = np.sqrt(s1)
        conv = rn < stop_after
        alphas.append(alpha[..., 0, :])
        betas.append(beta[..., 0, :])
        rns.append(rn[..., 0, :])
    return x * nrm, np.stack(alphas), np.stack(betas), np.stack(rns)


def tridiag_from(alphas, betas):
    """The reference's recurrence (linear_cg.py:311-332) from recorded alpha / beta: [T, T, B, c]."""
    T = alphas.shape[0]
    t_mat = np.zeros((T, T) + alphas.shape[1:], dtype=alphas.dtype)
    for k in range(T):
        ar = 1 / alphas[k]
        if k == 0:
            t_mat[0, 0] = ar
        else:
            t_mat[k, k] = ar + betas[k - 1] / alphas[k - 1]
            off = np.sqrt(betas[k - 1]) / alphas[k - 1]
            t_mat[k, k - 1] = off
            t_mat[k - 1, k] = off
    return t_mat


def main():
    B, N, R, c, k = 4, 8192, 32, 17, 15
    C, d, rhs = cases.lowrank_diag(99, B, N, R, c)
    rhs[..., :16] /= np.sqrt(np.sum(rhs[..., :16] ** 2, axis=-2, keepdims=True))
    L, _ = orc.pivoted_cholesky(orc.LowRankRowSource(C), k)
    pre = orc.Preconditioner(L, d)
    Q = pre.Q.astype(f32)
    mm = lambda v: orc.matvec_lowrank_diag(C, d, v)  # noqa: E731
    x32, t32, info = orc.linear_cg(mm, rhs, n_tridiag=16, tolerance=1e-4, preconditioner=pre.apply)
    C64, d64, rhs64 = C.astype(np.float64), d.astype(np.float64), rhs.astype(np.float64)
    pre64 = orc.Preconditioner(L.astype(np.float64), d64)
    mm64 = lambda v: orc.matvec_lowrank_diag(C64, d64, v)  # noqa: E731
    x64, t64, _ = orc.linear_cg(mm64, rhs64, n_tridiag=16, tolerance=1e-4, preconditioner=pre64.apply)
    print("oracle iterations", info.iterations)
    xs, al, be, rn = single_reduction_cg(C, d, Q, rhs, info.iterations)

    def rel(a, b):
        return float(np.max(np.linalg.norm(a - b, axis=-2) / np.linalg.norm(b, axis=-2)))

    print("x: oracle32 vs fp64      ", rel(x32, x64))
    print("x: single-red32 vs fp64  ", rel(xs, x64))
    print("x: single-red32 vs orc32 ", rel(xs, x32))
    ts = tridiag_from(al[:20, :, :16], be[:20, :, :16])  # [T,T,B,16]
    ts = np.transpose(ts, (3, 2, 0, 1))
    T = min(ts.shape[-1], t32.shape[-1])
    ev_s = np.linalg.eigvalsh(ts[..., :T, :T].astype(np.float64))
    ev_o = np.linalg.eigvalsh(t32[..., :T, :T].astype(np.float64))
    ev_6 = np.linalg.eigvalsh(t64[..., :T, :T])
    print("tridiag eig: orc32 vs fp64 ", float(np.abs(ev_o - ev_6).max()))
    print("tridiag eig: sr32 vs fp64  ", float(np.abs(ev_s - ev_6).max()))
    print("final resid norms (sr)", rn[-1].max(), " last alpha", al[-1].mean(), "beta", be[-1].mean())
    # logdet estimate differences (SLQ with weights): sum_j w_j log(lam_j), here just compare sum log eig of leading blocks
    for name, tm in (("orc32", t32), ("sr32", ts), ("fp64", t64)):
        ev, evec = np.linalg.eigh(tm[..., :T, :T].astype(np.float64))
        est = (evec[..., 0, :] ** 2 * np.log(ev)).sum(-1).mean(0) * N
        print("slq logdet part", name, est)


if __name__ == "__main__":
    main()
