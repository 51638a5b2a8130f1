"""Where does the time of a small-batch (cfg2: B=64) end-to-end solve go: device kernels vs host calls."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
B, N, R = 64, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d)
def e2e(timers=None):
    t = [time.perf_counter()]
    L, _ = K.pivoted_cholesky(desc, 15, contiguous=False); t.append(time.perf_counter())
    pre = K.precond_build(L, d, False); t.append(time.perf_counter())
    r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4); t.append(time.perf_counter())
    if timers is not None:
        for i in range(3): timers[i] += t[i + 1] - t[i]
    return r
for _ in range(5): e2e()
torch.cuda.synchronize(); tm = [0.0, 0.0, 0.0]; n = 50; t0 = time.perf_counter()
for _ in range(n): e2e(tm)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / n
print(f"e2e {tot*1e6:.0f} us per call: host-side wall pivchol {tm[0]/n*1e6:.0f} (sync), precond_build {tm[1]/n*1e6:.0f} (async), cg {tm[2]/n*1e6:.0f} (sync)")
_hip.prof_enable(True); e2e(); torch.cuda.synchronize()
print("kernels:", {k: round(ms / c * 1e3, 1) for k, (c, ms) in sorted(_hip.prof_report().items())})
