"""Forward + backward of A.solve(b) and A.inv_quad_logdet(b) at the cfg3 batch."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
C = (torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5).requires_grad_(True)
d = (torch.rand(B, N, generator=g, device="cuda") + 0.5).requires_grad_(True)
rhs = torch.randn(B, N, 1, generator=g, device="cuda").requires_grad_(True)
def step_solve():
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(C), DiagLinearOperator(d))
    A.solve(rhs).sum().backward()
def step_iql():
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(C), DiagLinearOperator(d))
    iq, ld = A.inv_quad_logdet(rhs, logdet=True)
    (iq.sum() + ld.sum()).backward()
with settings.cg_tolerance(1e-4), settings.num_trace_samples(16):
    for name, fn in (("solve", step_solve), ("inv_quad_logdet", step_iql)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); print(f"{name}: forward + backward {B} x {N}: {(time.perf_counter()-t0)/3*1e3:.2f} ms")
        _hip.prof_enable(True); fn(); torch.cuda.synchronize()
        p = _hip.prof_report(); _hip.prof_enable(False)
        tot = sum(ms for _, ms in p.values())
        print(f"    (liblo_amd kernels: {tot:.2f} ms; the rest is ATen plumbing + host)")
        for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"    {k:20s} {c:3d} x {ms / c * 1e3:9.1f} us  = {ms:7.2f} ms")
