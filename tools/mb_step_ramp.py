"""Per-step host times of the headline solve right after start-up: is there a ramp (clocks, allocator, caches)?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from linear_operator_amd import kernels as K
dev = torch.device("cuda", 0)
Cm, d, rhs = bench.make_problem(dev, 1234)
desc = K.lowrank_diag_descriptor(Cm, d)
pre = bench.build_precond(desc, d)
bench._gc_off()
laps = []
for i in range(400):
    t0 = time.perf_counter(); K.cg_solve(desc, rhs, precond=pre, tolerance=bench.TOL); laps.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("first 30:", " ".join(f"{x:.3f}" for x in laps[:30]))
for a in (30, 60, 100, 200, 300):
    seg = sorted(laps[a:a + 30]); print(f"steps {a}..{a + 30}: median {seg[15]:.4f} min {seg[0]:.4f} max {seg[-1]:.4f}")
