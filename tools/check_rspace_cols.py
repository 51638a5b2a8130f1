"""Three-launch R-space CG for several columns (k_rs_part / k_rs_iter / k_rs_apply) against the lockstep + serial resident
kernels: solutions, tridiagonals, iteration counts, timing."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from linear_operator_amd import kernels as K
dev = torch.device("cuda")


def exact(C, d, rhs):
    C64, d64, r64 = (torch.from_numpy(a).double().cuda() for a in (C, d, rhs))
    Cd = C64 / d64.unsqueeze(-1)
    cap = torch.eye(C64.shape[-1], dtype=torch.float64, device="cuda") + C64.mT @ Cd
    return (r64 / d64.unsqueeze(-1) - Cd @ torch.linalg.solve(cap, C64.mT @ (r64 / d64.unsqueeze(-1))))


def rel(a, b):
    return float(((a.double() - b.double()).norm(dim=-2) / b.double().norm(dim=-2)).max())


def run(N, R, B, c, nt, dscale=1.0, doff=0.5, reps=10):
    C, d, rhs = cases.lowrank_diag(7700 + R + c, B, N, R, c)
    d = ((d - 0.5) * dscale + doff).astype(np.float32)
    if nt:
        rhs[..., :nt] /= np.linalg.norm(rhs[..., :nt], axis=-2, keepdims=True)
    Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
    desc = K.lowrank_diag_descriptor(Cd, dd)
    L, perm = K.pivoted_cholesky(desc, 15)
    pre = K.precond_build(L, dd, constant_diag=False, root=desc.A0, perm=perm)
    ex = exact(C, d, rhs)
    out = {}
    for name, env in (("rs", {}), ("old", {"LO_NO_RSPACE_COLS": "1"})):
        os.environ.pop("LO_NO_RSPACE_COLS", None)
        os.environ.update(env)
        for _ in range(2):
            res = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4, n_tridiag=nt)
        ran = K.cg_last_executed()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            K.cg_solve(desc, rd, precond=pre, tolerance=1e-4, n_tridiag=nt)
        torch.cuda.synchronize()
        out[name] = (res, (time.perf_counter() - t0) / reps * 1e3, ran)
    os.environ.pop("LO_NO_RSPACE_COLS", None)
    a, b = out["rs"][0], out["old"][0]
    tm = ""
    if nt:
        ta, tb = a.t_mat.double(), b.t_mat.double()
        m = min(ta.shape[-1], tb.shape[-1])
        tm = f" t_mat shapes {tuple(ta.shape)} {tuple(tb.shape)} max rel diff {float(((ta[..., :m, :m] - tb[..., :m, :m]).abs().amax((-1, -2)) / tb[..., :m, :m].abs().amax((-1, -2))).max()):.1e}"
    print(f"N={N} R={R} B={B} c={c} nt={nt} d[{doff},{doff + dscale}]: rs {out['rs'][1]:.3f} ms ({out['rs'][2]['rspace']}, lean {out['rs'][2]['lean']}) "
          f"old {out['old'][1]:.3f} ms ({out['old'][2]['rspace']}) | iters {a.iterations}/{b.iterations} tol {a.tolerance_reached}/{b.tolerance_reached} "
          f"| err vs exact rs {rel(a.x, ex):.1e} old {rel(b.x, ex):.1e} | rs vs old {rel(a.x, b.x):.1e}{tm}")


if __name__ == "__main__":
    run(8192, 32, 64, 17, 16)
    run(8192, 32, 512, 17, 16, reps=5)
    run(8192, 32, 40, 3, 0)
    run(4096, 16, 24, 5, 4)
    run(5000, 8, 24, 2, 0)
    run(3000, 20, 16, 20, 16)
    run(16384, 32, 12, 17, 16)
    run(8192, 32, 24, 17, 16, dscale=0.1, doff=0.01)
    run(8192, 32, 24, 1, 1)
    run(8192, 32, 24, 32, 16)


def logdet_check(N=8192, R=32, B=16, P=16):
    C, d, rhs = cases.lowrank_diag(7900, B, N, R, P + 1)
    rhs[..., :P] /= np.linalg.norm(rhs[..., :P], axis=-2, keepdims=True)
    Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
    desc = K.lowrank_diag_descriptor(Cd, dd)
    L, perm = K.pivoted_cholesky(desc, 15)
    pre = K.precond_build(L, dd, constant_diag=False, root=desc.A0, perm=perm)
    C64, d64 = torch.from_numpy(C).double().cuda(), torch.from_numpy(d).double().cuda()
    cap = torch.eye(R, dtype=torch.float64, device="cuda") + C64.mT @ (C64 / d64.unsqueeze(-1))
    exact = torch.logdet(cap) + d64.log().sum(-1)
    for name, env in (("rs", {}), ("old", {"LO_NO_RSPACE_COLS": "1"})):
        os.environ.pop("LO_NO_RSPACE_COLS", None); os.environ.update(env)
        res = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4, n_tridiag=P)
        _, _, pinvk = K.tridiag_eigh_slq(res.t_mat, N)
        ld = (pinvk + pre.logdet).double()
        t = res.t_mat.double()
        print(name, "t_mat", tuple(t.shape), "logdet rel err vs exact (per member max)", float(((ld - exact).abs() / exact.abs()).max()),
              "first rows", t[0, 0].diagonal()[:6].tolist(), t[0, 0].diagonal(1)[:5].tolist())
    os.environ.pop("LO_NO_RSPACE_COLS", None)


if __name__ == "__main__":
    logdet_check()
