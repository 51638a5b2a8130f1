"""`LowRankRoot + Diag` (Woodbury closed form) at the cfg3 batch: solve + logdet."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip
from linear_operator_amd.operators import DiagLinearOperator, LowRankRootLinearOperator
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
C = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
def run():
    A = LowRankRootLinearOperator(C) + DiagLinearOperator(d)
    return A.solve(rhs), A.logdet()
run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): x, ld = run()
torch.cuda.synchronize(); print(f"Woodbury solve + logdet, {B} x {N} (R={R}): {(time.perf_counter()-t0)/3*1e3:.2f} ms")
_hip.prof_enable(True); run(); torch.cuda.synchronize()
for k, (c, ms) in sorted(_hip.prof_report().items()): print(f"    {k:20s} {c:4d} x {ms / c * 1e3:9.1f} us")
