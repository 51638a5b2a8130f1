#!/bin/bash
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^E " gpurun_out/pytest_gpu.log | head -8
python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from linear_operator_amd import _hip, kernels as K
g = torch.Generator(device="cuda"); g.manual_seed(1)
for N in (300, 515, 900, 1000):
    B, R = 1024, 32
    Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    d = torch.rand(B, N, generator=g, device="cuda") + 0.5
    rhs = torch.randn(B, N, 17, generator=g, device="cuda")
    desc = K.lowrank_diag_descriptor(Cm, d)
    def run():
        L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
        pre = K.precond_build(L, d, False, root=Cm, perm=perm)
        return K.cg_solve(desc, rhs, precond=pre, n_tridiag=16, tolerance=1e-4)
    run(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    _hip.prof_enable(True); run(); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    print(f"N={N}: factorise + build + 17-column CG with tridiagonals {dt*1e3:.2f} ms  [{', '.join(sorted(p, key=lambda k: -p[k][1])[:4])}]")
PY
