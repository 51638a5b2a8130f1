"""Resident / streaming CG over the shape range (c = 1, rank-15 preconditioner, 11 iterations at the floor): time per
solve and the equivalent streamed bytes per second -- looks for cliffs between the engines."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for R in (32, 16, 8):
    for N in (1024, 2048, 4096, 8192, 12000, 16384, 32768, 40000, 65536):
        B = max(8, min(1024, int(2 ** 32 / (N * R * 4) / 2)))
        Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
        d = torch.rand(B, N, generator=g, device=dev) + 0.5
        rhs = torch.randn(B, N, 1, generator=g, device=dev)
        desc = K.lowrank_diag_descriptor(Cm, d)
        L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
        pre = K.precond_build(L, d, False, root=Cm, perm=perm)
        for _ in range(2): r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        _hip.prof_enable(True); K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4); torch.cuda.synchronize()
        p = _hip.prof_report(); _hip.prof_enable(False)
        top = max(p, key=lambda k: p[k][1])
        per_it = 4.0 * B * N * (R + 16 + 10)  # streamed bytes of one iteration (C, Q, vectors once)
        print(f"R={R:2d} N={N:6d} B={B:4d}: {dt*1e3:8.3f} ms / solve, {r.iterations} it, {B * r.matvecs / dt / 1e6:6.2f} M member-matvecs/s, "
              f"equiv {per_it * r.iterations / dt / 1e12:5.1f} TB/s  [{top}]")
        del Cm, d, rhs, L, pre
