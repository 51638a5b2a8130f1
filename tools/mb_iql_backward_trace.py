"""One inv_quad_logdet forward + backward step at the cfg3 shape under the torch profiler: which ATen kernels the host
glue launches (name, calls, device time) next to the library's own."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from torch.profiler import profile, ProfilerActivity
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
C = (torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5).requires_grad_(True)
d = (torch.rand(B, N, generator=g, device="cuda") + 0.5).requires_grad_(True)
rhs = torch.randn(B, N, 1, generator=g, device="cuda").requires_grad_(True)
def step():
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(C), DiagLinearOperator(d))
    iq, ld = A.inv_quad_logdet(rhs, logdet=True)
    (iq.sum() + ld.sum()).backward()
with settings.cg_tolerance(1e-4), settings.num_trace_samples(16):
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60, max_shapes_column_width=70))
