"""R-space CG (k_cg_rspace) against the three-pass kernel, the w-recurrence kernel and the exact solution (fp64 Woodbury)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases
from linear_operator_amd import kernels as K

dev = torch.device("cuda")


def exact(C, d, rhs):
    C64, d64, r64 = (torch.from_numpy(a).double().cuda() for a in (C, d, rhs))
    Cd = C64 / d64.unsqueeze(-1)
    cap = torch.eye(C64.shape[-1], dtype=torch.float64, device="cuda") + C64.mT @ Cd
    return (r64 / d64.unsqueeze(-1) - Cd @ torch.linalg.solve(cap, C64.mT @ (r64 / d64.unsqueeze(-1))))


def rel(a, b):
    return float(((a.double() - b).norm(dim=-2) / b.norm(dim=-2)).max())


def run(N, R, B, dscale, doff, cscale, const=False, rank=15):
    C, d, rhs = cases.lowrank_diag(8800 + R, B, N, R, 1)
    C = (C * cscale).astype(np.float32)
    d = ((d - 0.5) * dscale + doff).astype(np.float32)
    if const:
        d = d[:, 0].copy()
    Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
    desc = K.lowrank_diag_descriptor(Cd, dd, const)
    L, perm = K.pivoted_cholesky(desc, rank)
    pre = K.precond_build(L, dd, constant_diag=const, root=desc.A0, perm=perm)
    assert pre.RS is not None
    ex = exact(C, d if not const else np.repeat(d[:, None], N, 1), rhs)
    out = {}
    for name, env in (("rspace", {}), ("wrec", {"LO_OC_NO_RSPACE": "1"}), ("3pass", {"LO_OC_NO_RSPACE": "1", "LO_OC_NO_WREC": "1"})):
        for k in ("LO_OC_NO_RSPACE", "LO_OC_NO_WREC"):
            os.environ.pop(k, None)
        os.environ.update(env)
        res = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
        ran = K.cg_last_executed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
        torch.cuda.synchronize()
        out[name] = (res, rel(res.x, ex), (time.perf_counter() - t0) / 20 * 1e3, ran["lean"], ran["serial_engine"])
    for k in ("LO_OC_NO_RSPACE", "LO_OC_NO_WREC"):
        os.environ.pop(k, None)
    r0 = out["rspace"][0]
    print(f"N={N} R={R} B={B} d in [{doff:g},{doff+dscale:g}] Cx{cscale} const={const}: iters {r0.iterations} tol_reached {r0.tolerance_reached} | "
          + " | ".join(f"{n}: err {e:.1e} {ms:.3f} ms lean={ln} eng={eng}" for n, (_, e, ms, ln, eng) in out.items())
          + f" | rspace vs 3pass {rel(r0.x, out['3pass'][0].x.double()):.1e} mean resid {r0.mean_residual:.2e} vs {out['3pass'][0].mean_residual:.2e}")
    # (ill-conditioned members: the fp64 R-space iteration converges like exact CG and stops at the floor where the fp32
    #  iterations go on -- iteration counts may differ there)


if __name__ == "__main__":
    run(8192, 32, 512, 1.0, 0.5, 1.0)
    run(8192, 32, 40, 1.0, 0.5, 1.0)
    run(8192, 32, 24, 0.1, 0.01, 1.0)
    run(2048, 32, 24, 0.01, 0.001, 1.0)
    run(4096, 16, 24, 1.0, 0.05, 1.0)
    run(5000, 8, 24, 1.0, 0.5, 1.0)
    run(8192, 32, 24, 1.0, 0.5, 10.0)
    run(16384, 32, 12, 1.0, 0.5, 1.0)
    run(1024, 32, 24, 10.0, 0.5, 1.0)
    run(40000, 32, 6, 1.0, 0.5, 1.0)
    run(8192, 32, 64, 1.0, 0.5, 1.0, const=True)
    run(3000, 20, 16, 1.0, 0.5, 1.0)
