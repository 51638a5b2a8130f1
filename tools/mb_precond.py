import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
desc = K.lowrank_diag_descriptor(Cm, d)
for _ in range(2):
    L, _ = K.pivoted_cholesky(desc, 15, contiguous=False); pre = K.precond_build(L, d, False)
torch.cuda.synchronize()
_hip.prof_enable(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    L, _ = K.pivoted_cholesky(desc, 15, contiguous=False); pre = K.precond_build(L, d, False)
e1.record(); torch.cuda.synchronize()
print("precond build total ms", e0.elapsed_time(e1) / 3)
for k, (c, ms) in sorted(_hip.prof_report().items()):
    print(f"  {k:18s} {c:4d} x {ms / c * 1e3:8.1f} us")
