"""Where the time of an API-level inv_quad_logdet forward + backward goes (ATen kernels included)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from torch.profiler import profile, ProfilerActivity
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
C = (torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5).requires_grad_(True)
d = (torch.rand(B, N, generator=g, device="cuda") + 0.5).requires_grad_(True)
rhs = torch.randn(B, N, 1, generator=g, device="cuda").requires_grad_(True)
def fwd():
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(C), DiagLinearOperator(d))
    iq, ld = A.inv_quad_logdet(rhs, logdet=True)
    return iq.sum() + ld.sum()
with settings.cg_tolerance(1e-4), settings.num_trace_samples(16):
    fwd().backward(); torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter(); l = fwd(); torch.cuda.synchronize(); t1 = time.perf_counter()
        l.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"forward {1e3*(t1-t0):.2f} ms, backward {1e3*(t2-t1):.2f} ms")
    if len(sys.argv) > 1 and sys.argv[1] == "forward":
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            fwd(); torch.cuda.synchronize()
    else:
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            fwd().backward(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
