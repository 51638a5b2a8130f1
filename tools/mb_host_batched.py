"""Batched low-rank + diag problems through the host API over the member size: inv_quad_logdet forward / backward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
g = torch.Generator(device="cuda"); g.manual_seed(1)
with settings.cg_tolerance(1e-4), settings.num_trace_samples(16):
    for N in (1024, 2048, 4096, 8192):
        B, R = 512, 32
        C = (torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5).requires_grad_(True)
        d = (torch.rand(B, N, generator=g, device="cuda") + 0.5).requires_grad_(True)
        rhs = torch.randn(B, N, 1, generator=g, device="cuda")
        def fwd():
            A = AddedDiagLinearOperator(LowRankRootLinearOperator(C), DiagLinearOperator(d))
            iq, ld = A.inv_quad_logdet(rhs, logdet=True)
            return iq.sum() + ld.sum()
        fwd().backward(); torch.cuda.synchronize()
        tf = tb = 0.0
        for _ in range(3):
            t0 = time.perf_counter(); l = fwd(); torch.cuda.synchronize(); t1 = time.perf_counter()
            l.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
            tf += t1 - t0; tb += t2 - t1
        print(f"B={B} N={N:5d}: forward {tf/3*1e3:6.2f} ms, backward {tb/3*1e3:6.2f} ms")
