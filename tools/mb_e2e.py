"""End-to-end solve (pivoted Cholesky + Woodbury build + CG) at the headline shape; run under rocprofv3 to see
every kernel including the torch glue."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d)
def e2e():
    L, perm = K.pivoted_cholesky(desc, 15, contiguous=False)
    pre = K.precond_build(L, d, False, root=Cm, perm=perm, need_q=False)  # root form only (what the resident CG needs)
    return K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
for _ in range(3): e2e()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): res = e2e()
torch.cuda.synchronize(); print("e2e ms", (time.perf_counter() - t0) / 5 * 1e3, "iters", res.iterations)
