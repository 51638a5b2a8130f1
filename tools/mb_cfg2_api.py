"""cfg2 through the operator API (A.solve, one fused launch): wall time per call against the kernel, and the Python
profile of the host side (where the ~100 us around a 144 us kernel go)."""
import os, sys, time, cProfile, pstats, io, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
try:
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
except Exception:  # noqa: BLE001
    clear_preconditioner_memo = lambda: None  # noqa: E731
B, N, R = int(os.environ.get("B", 64)), 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
def solve():
    clear_preconditioner_memo()
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
    return A.solve(rhs)
with settings.cg_tolerance(1e-4), torch.no_grad():
    for _ in range(10): solve()
    torch.cuda.synchronize(); n = 200; t0 = time.perf_counter()
    for _ in range(n): solve()
    torch.cuda.synchronize(); print(f"A.solve: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call")
    _hip.prof_enable(True); solve(); torch.cuda.synchronize()
    print("kernels:", {k: round(ms / c * 1e3, 1) for k, (c, ms) in sorted(_hip.prof_report().items())}); _hip.prof_enable(False)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): solve()
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
