"""cfg3 forward + backward through the operator API (bench.py's train_step), timed step by step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings as lo_settings, kernels as K, _hip
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(77)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
y = torch.randn(B, N, 1, generator=g, device=dev)
Cg, dg = Cm.clone().requires_grad_(True), d.clone().requires_grad_(True)
def train_step():
    clear_preconditioner_memo()
    Cg.grad = dg.grad = None
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
    iq, ld = A.inv_quad_logdet(y, logdet=True)
    (iq.sum() + ld.sum()).backward()
    return iq
with lo_settings.cg_tolerance(1e-4), lo_settings.num_trace_samples(16):
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); train_step(); torch.cuda.synchronize()
        print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    if len(sys.argv) > 1:
        _hip.prof_enable(True); train_step(); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
        for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:14]: print(f"    {k:24s} {c:5d} x {ms / c * 1e3:9.1f} us  total {ms:8.2f} ms")
