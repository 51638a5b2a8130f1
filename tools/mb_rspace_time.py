import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
rhs = torch.randn(B, N, 1, generator=g, device=dev)
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)
for _ in range(3): K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
torch.cuda.synchronize(); print("ms/solve", (time.perf_counter() - t0) / 50 * 1e3, K.cg_last_executed())
