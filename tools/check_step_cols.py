"""k_cg_step_cols (csrc/lo_cg_step_cols.hip) against the multi-launch streaming iteration (LO_NO_STEP_COLS=1) over shapes:
dense / Kronecker operators, with and without the Woodbury preconditioner, 1 .. 32 columns, tridiagonals, batches
larger than the number of resident groups.  Prints the worst differences and the time per solve of both paths."""
import os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K

dev = torch.device("cuda")


def solve(desc, rhs, pre, nt, tol, max_iter=1000):
    t0 = time.perf_counter()
    res = K.cg_solve(desc, rhs, precond=pre, n_tridiag=nt, tolerance=tol, max_iter=max_iter)
    torch.cuda.synchronize()
    return res, time.perf_counter() - t0


def cg64(Kd, d, rhs, pre, iters, nt):
    """the iteration of linear_cg.py:245-332 in float64 with the same preconditioner (z = r/d - Q Q^T r): the exact
    CG coefficients both fp32 paths approximate"""
    A = Kd.double(); dd = d.double(); b = rhs.double()
    nrm = b.norm(dim=-2, keepdim=True); r = b / nrm; x = torch.zeros_like(r)
    if pre is not None:
        Q = pre.Q.double(); di = pre.dinv.double().unsqueeze(-1)
        prec = lambda v: v * di - Q @ (Q.mT @ v)
    else:
        prec = lambda v: v
    z = prec(r); p = z.clone(); rz = (r * z).sum(-2, keepdim=True)
    T = torch.zeros(nt, rhs.shape[0], iters, iters, dtype=torch.float64, device=rhs.device)
    pa = pb = None
    for k in range(iters):
        Ap = A @ p + dd.unsqueeze(-1) * p
        al = rz / (p * Ap).sum(-2, keepdim=True)
        x = x + al * p; r = r - al * Ap
        z = prec(r); rzn = (r * z).sum(-2, keepdim=True); be = rzn / rz; rz = rzn
        p = z + be * p
        ar = 1.0 / al[:, 0, :nt]  # [B, nt]
        if k == 0:
            T[:, :, 0, 0] = ar.T
        else:
            T[:, :, k, k] = (ar + pb * pa).T
            off = (pb.sqrt() * pa).T
            T[:, :, k, k - 1] = off; T[:, :, k - 1, k] = off
        pa, pb = ar, be[:, 0, :nt]
    return x * nrm, T


def case(kind, B, N, c, k, nt, seed, tol=1e-4):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    if kind == "dense":
        R_ = max(8, min(N // 4, 512))
        X = torch.randn(B, N, R_, generator=g, device=dev) / (R_ ** 0.5)
        Kd = X @ X.mT; del X
        d = torch.rand(B, N, generator=g, device=dev) * 0.5 + 0.25
        desc = K.dense_diag_descriptor(Kd, d)
        pdesc, pd, const = desc, d, False
    else:
        n1 = n2 = int(round(N ** 0.5)); N = n1 * n2
        X1 = torch.randn(B, n1, n1, generator=g, device=dev) / n1 ** 0.5; X2 = torch.randn(B, n2, n2, generator=g, device=dev) / n2 ** 0.5
        K1 = X1 @ X1.mT + 0.1 * torch.eye(n1, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n2, device=dev)
        sig = torch.full((B,), 0.05, device=dev)
        desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
        pdesc, pd, const = desc.without_diag(), sig, True
    rhs = torch.randn(B, N, c, generator=g, device=dev)
    pre = None
    if k:
        L, perm = K.pivoted_cholesky(pdesc, k)
        pre = K.precond_build(L, pd, const)
    out = {}
    for mode in ("cols", "multi"):
        if mode == "multi":
            os.environ["LO_NO_STEP_COLS"] = "1"
        else:
            os.environ.pop("LO_NO_STEP_COLS", None)
        solve(desc, rhs, pre, nt, tol)
        res, dt = solve(desc, rhs, pre, nt, tol)
        plan = K.cg_last_executed()
        out[mode] = (res, dt, plan)
    os.environ.pop("LO_NO_STEP_COLS", None)
    a, b = out["cols"][0], out["multi"][0]
    xe = ((a.x - b.x).norm(dim=-2) / b.x.norm(dim=-2)).max().item()
    te = 0.0
    if nt:
        m = min(a.t_mat.shape[-1], b.t_mat.shape[-1])
        te = ((a.t_mat[..., :m, :m] - b.t_mat[..., :m, :m]).abs().amax() / b.t_mat.abs().amax()).item()
    t64 = ""
    t_ok = True
    if nt and kind == "dense" and B * N * N <= (1 << 30):
        _, T64 = cg64(Kd, d, rhs, pre, a.t_mat.shape[-1], nt)
        m = a.t_mat.shape[-1]
        for kk_ in (5, 10, m):
            kk_ = min(kk_, m)
            sc = T64[..., :kk_, :kk_].abs().amax()
            ea = ((a.t_mat[..., :kk_, :kk_].double() - T64[..., :kk_, :kk_]).abs().amax() / sc).item()
            eb = ((b.t_mat[..., :kk_, :kk_].double() - T64[..., :kk_, :kk_]).abs().amax() / sc).item()
            t64 += f" [{kk_}: {ea:.1e} / {eb:.1e}]"
            t_ok = t_ok and ea <= max(3 * eb, 2e-6)  # as close to the exact coefficients as the multi-launch path
    # residual of the new path against the operator itself
    if kind == "dense":
        Ax = Kd @ a.x + d.unsqueeze(-1) * a.x
    else:
        Ax = torch.einsum("bij,bjkc->bikc", K1, torch.einsum("bkl,bjlc->bjkc", K2, a.x.reshape(B, n1, n2, c))).reshape(B, N, c) + 0.05 * a.x
    rres = ((Ax - rhs).norm(dim=-2) / rhs.norm(dim=-2)).max().item()
    stream = out["cols"][2]
    print(f"{kind:5s} B={B:4d} N={N:6d} c={c:2d} k={k:2d} nt={nt:2d}: iters {a.iterations}/{b.iterations}  x diff {xe:.2e}  "
          f"t_mat diff {te:.2e}  residual {rres:.2e}  {stream['streaming_precond']}  {out['cols'][1]*1e3:8.2f} ms vs {out['multi'][1]*1e3:8.2f} ms"
          + (f"\n        t_mat vs fp64, leading blocks (one launch / multi-launch): {t64}" if t64 else ""))
    ok = a.iterations == b.iterations and xe < 2e-4 and t_ok and rres < 5e-3
    return ok


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cases = [
        ("dense", 1, 100, 3, 0, 0), ("dense", 3, 256, 2, 0, 2), ("dense", 2, 700, 5, 3, 4), ("dense", 1, 1000, 11, 7, 10),
        ("dense", 1, 4000, 11, 15, 10), ("dense", 2, 4096, 16, 15, 16), ("dense", 3, 3001, 17, 15, 16),
        ("dense", 2, 5000, 32, 10, 8), ("dense", 1, 2000, 1, 15, 0), ("dense", 600, 300, 4, 5, 3), ("dense", 70, 1100, 3, 0, 0),
        ("kron", 4, 48 * 48, 3, 15, 0), ("kron", 2, 128 * 128, 5, 15, 4), ("kron", 40, 64 * 64, 2, 0, 0),
    ]
    if not quick:
        cases += [("dense", 4, 16384, 17, 15, 16), ("dense", 2, 12000, 11, 15, 10), ("dense", 10, 8192, 20, 12, 16),
                  ("dense", 1000, 300, 11, 5, 10), ("dense", 200, 1000, 11, 7, 10), ("dense", 64, 4096, 17, 15, 16),
                  ("dense", 24, 16384, 17, 15, 16), ("dense", 300, 600, 8, 0, 0), ("dense", 2000, 128, 9, 0, 4)]
    bad = 0
    for i, (kind, B, N, c, k, nt) in enumerate(cases):
        if not case(kind, B, N, c, k, nt, 100 + i):
            bad += 1
            print("   ^^^ MISMATCH")
    print("OK" if bad == 0 else f"{bad} FAILED")
    sys.exit(1 if bad else 0)
