"""Pivoted Cholesky + preconditioner build + CG against the preconditioner rank (512 x 8192, R = 32)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) * 0.1 + 0.01
rhs = torch.randn(B, N, 1, generator=g, device=dev)
desc = K.lowrank_diag_descriptor(Cm, d)
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
for k in (5, 10, 15, 16, 17, 24, 32, 40):
    tp, (L, perm) = t(lambda: K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), k, contiguous=False))
    tb, pre = t(lambda: K.precond_build(L, d, False, root=Cm, perm=perm))
    tc, r = t(lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4))
    print(f"rank {k:2d} (got {L.shape[-2] if L.shape[-1] == N else L.shape[-1]}): pivoted Cholesky {tp:6.2f} ms, build {tb:5.2f} ms, CG {tc:5.2f} ms ({r.iterations} it)")
