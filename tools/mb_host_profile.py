"""Where the host time of one single-column solve goes (cProfile over 3000 solves of the headline shape)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases
from linear_operator_amd import kernels as K

dev = torch.device("cuda")
C, d, rhs = cases.lowrank_diag(8832, 512, 8192, 32, 1)
Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
desc = K.lowrank_diag_descriptor(Cd, dd, False)
L, perm = K.pivoted_cholesky(desc, 15)
pre = K.precond_build(L, dd, constant_diag=False, root=desc.A0, perm=perm)
pre.ensure_eigform()
for _ in range(200):
    K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
