"""Host-side overhead of A.solve(rhs) on the fused path: cProfile over many solves of a SMALL batch (kernel time small)."""
import cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K, settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
B, N, R = int(os.environ.get("FU_B", 64)), 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
desc = K.lowrank_diag_descriptor(Cm, d)
def api():
    clear_preconditioner_memo()
    return A.solve(rhs)
def kern():
    return K.solve_fused(desc, rhs, 15, 1e-3, tolerance=1e-4)
with settings.cg_tolerance(1e-4):
    for fn in (api, kern):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): fn()
        torch.cuda.synchronize(); print(fn.__name__, "ms", (time.perf_counter() - t0) / 100 * 1e3)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): api()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
