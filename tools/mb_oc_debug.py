"""Phase timers of the resident CG kernel: run with LO_OC_DEBUG=<member index> (prints 100 MHz tick counts)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = int(os.environ.get('OC_B', 512)), int(os.environ.get('OC_N', 8192)), 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
rhs = torch.randn(B, N, 1, generator=g, device=dev)
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)  # Q form + root form
for _ in range(3): K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
