"""Resident against streaming CG for N beyond the headline shape: 128 x (32768^2 low-rank(32) + diag), 1 column."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for B, N, R in ((128, 32768, 32), (256, 16384, 32)):
    Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    rhs = torch.randn(B, N, 1, generator=g, device=dev)
    desc = K.lowrank_diag_descriptor(Cm, d)
    for on in (True, False):
        K.set_onchip_cg(on)
        K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15); torch.cuda.synchronize()
        t0 = time.perf_counter(); L, _ = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15); torch.cuda.synchronize()
        print(f"B={B} N={N}: resident={on}: pivoted Cholesky (rank 15) {1e3*(time.perf_counter()-t0):.2f} ms")
    pre = K.precond_build(L, d, False)
    for on in (True, False):
        K.set_onchip_cg(on)
        K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
        print(f"B={B} N={N}: resident={on}: {t*1e3:.2f} ms per solve batch, {r.iterations} iterations, {B*r.matvecs/t/1e6:.2f} M member-matvecs/s")
    K.set_onchip_cg(True)
    del Cm, d, rhs, L, pre
