"""Shifted MINRES / contour integral quadrature at the cfg3 batch: 512 x 8192 (R = 32), 1 column, 15 + 1 shifts."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K, settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
rhs = torch.randn(B, N, 1, generator=g, device=dev)
desc = K.lowrank_diag_descriptor(Cm, d)
shifts = torch.linspace(0.0, 30.0, 16, device=dev)
def run(): return K.minres_solve(desc, rhs, shifts, max_iter=1000)
run(); torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
per_it = t / r.iterations
alg = (3 + 4 * 16) * N * 4 * B + 4 * (2 * N * R + 3 * N) * B   # MINRES vectors + one low-rank matvec
print(f"MINRES 16 shifts: {t*1e3:.2f} ms, {r.iterations} iterations, {per_it*1e6:.0f} us/iteration, "
      f"{alg/per_it/1e9:.0f} GB/s algorithmic, conv {r.conv:.2e}")
_hip.prof_enable(True); run(); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:6]: print(f"    {k:20s} {c:5d} x {ms / c * 1e3:9.1f} us  total {ms:8.2f} ms")
A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
with settings.min_preconditioning_size(10 ** 9):
    f = lambda: A.sqrt_inv_matmul(rhs)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); x = f(); torch.cuda.synchronize()
    print(f"sqrt_inv_matmul (CIQ, 15 nodes, no preconditioner): {(time.perf_counter()-t0)*1e3:.2f} ms")
