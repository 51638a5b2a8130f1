"""Host overhead of the headline step: K.cg_solve wall time per call vs the resident kernel's HIP-event time."""
import cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(desc, 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)
step = lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50
K._hip.prof_enable(True)
for _ in range(10): step()
torch.cuda.synchronize()
prof = K._hip.prof_report(); K._hip.prof_enable(False)
print("step wall us", wall * 1e6, "kernels", {k: round(v[1] / v[0] * 1e3, 1) for k, v in prof.items()})
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
