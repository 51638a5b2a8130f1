import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import cases
from linear_operator_amd import kernels as K
dev = torch.device("cuda")
for (N, R, B, ds, do) in ((8192, 32, 512, 1.0, 0.5), (2048, 32, 24, 0.01, 0.001)):
    C, d, rhs = cases.lowrank_diag(8800 + R, B, N, R, 1)
    d = ((d - 0.5) * ds + do).astype(np.float32)
    Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
    desc = K.lowrank_diag_descriptor(Cd, dd)
    L, perm = K.pivoted_cholesky(desc, 15)
    pre = K.precond_build(L, dd, constant_diag=False, root=desc.A0, perm=perm)
    ts = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(N, B, ["%.3f" % t for t in ts], K.cg_last_executed()["lean"], r.mean_residual)
