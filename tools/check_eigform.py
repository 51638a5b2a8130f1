"""Diagonal form of the R-space CG (lo_precond_eigform_f32 + k_cg_rspace<.., true>) against the dense R-space kernel,
the three-pass kernel and the exact solution (fp64 Woodbury); the form itself against its defining identities."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases
from linear_operator_amd import kernels as K
from check_rspace import exact, rel

dev = torch.device("cuda")


def form_identities(pre, R):
    """TinT E Tin = Lam^-1 ... checked through what the iteration relies on:
    (1) Ep = E^+ on the kept directions: E Ep E = E;  (2) Tin^T (E + E^2) Tin = Lam and Tin^T (E - E F E)^+... = I via
    P_U: Tin^T (E + E M E) Tin = I  <=>  W^T P_U W = I.  With only E, F at hand: W^T A_U W = Lam reads Tin^T (E + E E) Tin = Lam,
    and P^-1 on span: Tin^T E (I - F E) ... = Tin^T (E - E F E) Tin = W^T P_U^-1... not the identity; use instead
    Tu^T (E - E F E) Tu = I (Tu = V S^-1 W^-T, W^-1 P_U^-1 W^-T = I)."""
    RS, D = pre.RS.double(), pre.RSD.double()
    E, F = RS[:, 0, :R, :R], RS[:, 4, :R, :R]
    TinT, Ep, TuT, Tin = D[:, 0, :R, :R], D[:, 1, :R, :R], D[:, 2, :R, :R], D[:, 4, :R, :R]
    lam = D[:, 5, 0, :R]
    sc = E.abs().amax((-1, -2), keepdim=True)
    e1 = ((E @ Ep @ E - E).abs() / sc).max().item()
    keep = (Tin.abs().amax(-2) > 0)
    A_u = TinT @ (E + E @ E) @ Tin
    e2 = (A_u - torch.diag_embed(lam * keep.double())).abs().max().item() / lam.abs().max().item()
    P_u = TuT @ (E - E @ F @ E) @ TuT.mT
    eye = torch.diag_embed(keep.double())
    e3 = (P_u - eye).abs().max().item()
    e4 = (TinT - Tin.mT).abs().max().item()
    st = D[:, 5, 1, :4]
    return e1, e2, e3, e4, st[:, 1].max().item(), st[:, 2].max().item(), st[:, 3].min().item(), lam.min().item(), lam.max().item()


def run(N, R, B, dscale, doff, cscale, const=False, rank=15, dup=False, reps=20):
    C, d, rhs = cases.lowrank_diag(8800 + R, B, N, R, 1)
    C = (C * cscale).astype(np.float32)
    if dup:
        C[..., R // 2:] = C[..., :R - R // 2]
    d = ((d - 0.5) * dscale + doff).astype(np.float32)
    if const:
        d = d[:, 0].copy()
    Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
    desc = K.lowrank_diag_descriptor(Cd, dd, const)
    L, perm = K.pivoted_cholesky(desc, rank)
    pre = K.precond_build(L, dd, constant_diag=const, root=desc.A0, perm=perm)
    assert pre.RS is not None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pre.ensure_eigform()
    torch.cuda.synchronize()
    t_form = (time.perf_counter() - t0) * 1e3
    assert torch.is_tensor(pre.RSD), "eigform unusable"
    ids = form_identities(pre, R)
    ex = exact(C, d if not const else np.repeat(d[:, None], N, 1), rhs)
    out = {}
    for name, env in (("diag", {}), ("rspace", {"LO_RS_NO_DIAG": "1"}), ("3pass", {"LO_OC_NO_RSPACE": "1", "LO_OC_NO_WREC": "1"})):
        for k in ("LO_OC_NO_RSPACE", "LO_OC_NO_WREC", "LO_RS_NO_DIAG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        res = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
        ran = K.cg_last_executed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
        torch.cuda.synchronize()
        out[name] = (res, rel(res.x, ex), (time.perf_counter() - t0) / reps * 1e3, ran)
    for k in ("LO_OC_NO_RSPACE", "LO_OC_NO_WREC", "LO_RS_NO_DIAG"):
        os.environ.pop(k, None)
    r0, r1 = out["diag"][0], out["rspace"][0]
    assert out["diag"][3]["rspace_diag"] and not out["rspace"][3]["rspace_diag"], (out["diag"][3], out["rspace"][3])
    print(f"N={N} R={R} B={B} d in [{doff:g},{doff+dscale:g}] Cx{cscale} const={const} dup={dup}: form {t_form:.3f} ms "
          f"(E E+ E {ids[0]:.1e}, A_U {ids[1]:.1e}, P_U {ids[2]:.1e}, T {ids[3]:.1e}, sweeps {ids[4]:.0f}/{ids[5]:.0f}, min rank {ids[6]:.0f}, lam [{ids[7]:.3g}, {ids[8]:.3g}])\n"
          f"   iters {r0.iterations}/{r1.iterations} tol {r0.tolerance_reached}/{r1.tolerance_reached} mean resid {r0.mean_residual:.3e}/{r1.mean_residual:.3e} | "
          + " | ".join(f"{n}: err {e:.1e} {ms:.3f} ms" for n, (_, e, ms, _r) in out.items())
          + f" | diag vs rspace {rel(r0.x, r1.x.double()):.1e}")


if __name__ == "__main__":
    run(8192, 32, 512, 1.0, 0.5, 1.0, reps=200)
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        run(8192, 32, 40, 1.0, 0.5, 1.0)
        sys.exit(0)
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
    run(4096, 32, 24, 1.0, 0.5, 1.0, dup=True)
