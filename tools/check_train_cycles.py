"""Does a training step (A.inv_quad_logdet forward + backward through the operator API) leave cyclic garbage that keeps
device memory alive until the cyclic collector runs?  Memory after each step with the collector off, then what a
collection frees and which object types sat in the cycles."""
import gc, os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
B, N, R = int(os.environ.get("B", 64)), 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cg = (torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5).requires_grad_(True)
dg = (torch.rand(B, N, generator=g, device="cuda") + 0.5).requires_grad_(True)
y = torch.randn(B, N, 1, generator=g, device="cuda")
def step():
    clear_preconditioner_memo()
    Cg.grad = dg.grad = None
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
    iq, ld = A.inv_quad_logdet(y, logdet=True)
    (iq.sum() + ld.sum()).backward()
gc.collect(); gc.disable()
with settings.cg_tolerance(1e-4), settings.num_trace_samples(16):
    step(); torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for i in range(4):
        step(); torch.cuda.synchronize()
        print(f"after step {i}: {(torch.cuda.memory_allocated() - base) / 1e6:10.1f} MB above the first step")
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    print(f"collector found {n} objects; memory now {(torch.cuda.memory_allocated() - base) / 1e6:.1f} MB above the first step")
    cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
    print(cnt.most_common(25))
    for o in gc.garbage:
        if type(o).__name__ in ("function", "cell") :
            continue
    fn = [o for o in gc.garbage if type(o).__name__ == "function"]
    print([f"{f.__module__}.{f.__qualname__}" for f in fn][:40])
