"""End-to-end solve at the headline shape, the round-3 way: ONE resident launch (lo_solve_fused_f32) -- through the
C-ABI wrapper and through the operator API (A.solve, memo cleared) -- next to the round-2 three-launch path.
Run under rocprofv3 to see every kernel including the torch glue."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K, settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d)
A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
def three():
    L, perm = K.pivoted_cholesky(desc, 15, contiguous=False)
    pre = K.precond_build(L, d, False, root=Cm, perm=perm, need_q=False)  # root form only (what the resident CG needs)
    return K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4).x
def fused():
    return K.solve_fused(desc, rhs, 15, 1e-3, tolerance=1e-4).cg.x
def api():
    clear_preconditioner_memo()
    return A.solve(rhs)
with settings.cg_tolerance(1e-4):
    for name, fn in (("three-launch (C ABI wrappers)", three), ("fused, one launch (C ABI wrapper)", fused),
                     ("fused through A.solve (operator API)", api)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): x = fn()
        torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
