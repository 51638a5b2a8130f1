"""Fused end-to-end solve (lo_solve_fused_f32) against the three-launch path on the same inputs: solution, root-form
matrices, logdet P, pivots; then timings.  FU_B / FU_N / FU_R / FU_C / FU_CONST select the shape."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
B, N, R, c = (int(os.environ.get(k, v)) for k, v in (("FU_B", 512), ("FU_N", 8192), ("FU_R", 32), ("FU_C", 1)))
const = bool(int(os.environ.get("FU_CONST", 0)))
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = (torch.rand(B, generator=g, device="cuda") + 0.5) if const else (torch.rand(B, N, generator=g, device="cuda") + 0.5)
rhs = torch.randn(B, N, c, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d, const_diag=const)
def three():
    L, perm = K.pivoted_cholesky(desc, 15, contiguous=False)
    pre = K.precond_build(L, d, const, root=Cm, perm=perm, need_q=False)
    return K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4), pre, perm
def fused():
    return K.solve_fused(desc, rhs, 15, 1e-3, tolerance=1e-4)
r3, pre3, perm3 = three()
rf = fused()
assert rf is not None, "fused path declined"
rel = ((rf.cg.x - r3.x).flatten(1).norm(dim=1) / r3.x.flatten(1).norm(dim=1)).max().item()
print("iters", rf.cg.iterations, r3.iterations, "tol", rf.cg.tolerance_reached, "mean resid", rf.cg.mean_residual, r3.mean_residual)
print("x rel diff fused vs three-launch:", rel)
print("F rel", ((rf.precond.F - pre3.F).norm() / pre3.F.norm()).item(), "EF rel", ((rf.precond.EF - pre3.EF).norm() / pre3.EF.norm()).item(),
      "E rel", ((rf.precond.E - pre3.E).norm() / pre3.E.norm()).item())
print("logdet max abs diff", (rf.precond.logdet.reshape(-1) - pre3.logdet.reshape(-1)).abs().max().item(), "of", pre3.logdet.abs().max().item())
print("dinv equal", torch.equal(rf.precond.dinv, pre3.dinv))
print("perm equal", torch.equal(rf.permutation(N), perm3.reshape(B, N)))
# exact solution (fp64 Woodbury)
C64, r64 = Cm.double(), rhs.double()
d64 = (d.double().reshape(B, 1, 1).expand(B, N, 1) if const else d.double().unsqueeze(-1))
Cd = C64 / d64
cap = torch.eye(R, device="cuda", dtype=torch.float64) + C64.mT @ Cd
xs = r64 / d64 - Cd @ torch.linalg.solve(cap, C64.mT @ (r64 / d64))
for nm, x in (("fused", rf.cg.x), ("three", r3.x)):
    print(nm, "max rel err vs fp64 closed form", ((x.double() - xs).flatten(1).norm(dim=1) / xs.flatten(1).norm(dim=1)).max().item())
for nm, fn in (("three-launch", three), ("fused", fused)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(nm, "ms", (time.perf_counter() - t0) / 10 * 1e3)
again = K.cg_solve(desc, rhs, precond=rf.precond, tolerance=1e-4)
dx = (again.x - rf.cg.x).abs().max().item()
print("fused x vs root-form CG kernel with the fused launch's own preconditioner: bit-equal", torch.equal(again.x, rf.cg.x), "max abs diff", dx)
rf2 = fused()
print("fused twice bit-equal", torch.equal(rf2.cg.x, rf.cg.x))
K._hip.prof_enable(True)
for _ in range(5): fused()
torch.cuda.synchronize()
print("prof fused:", {k: round(v[1] / v[0] * 1e3, 1) for k, v in K._hip.prof_report().items()}, "us per launch")
for _ in range(5): three()
torch.cuda.synchronize()
print("prof three:", {k: round(v[1] / v[0] * 1e3, 1) for k, v in K._hip.prof_report().items()}, "us per launch")
K._hip.prof_enable(False)
