"""LO_OC_DEBUG stamps of the single-column R-space kernel (dense and diagonal form) at the headline shape."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases
from linear_operator_amd import kernels as K

dev = torch.device("cuda")
C, d, rhs = cases.lowrank_diag(8832, 512, 8192, 32, 1)
Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
desc = K.lowrank_diag_descriptor(Cd, dd, False)
L, perm = K.pivoted_cholesky(desc, 15)
pre = K.precond_build(L, dd, constant_diag=False, root=desc.A0, perm=perm)
pre.ensure_eigform()
for _ in range(50):
    K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
torch.cuda.synchronize()
for mode in ("diag", "dense"):
    os.environ.pop("LO_RS_NO_DIAG", None)
    if mode == "dense":
        os.environ["LO_RS_NO_DIAG"] = "1"
    for m in (0, 200, 450):
        os.environ["LO_OC_DEBUG"] = str(m)
        sys.stderr.write(f"--- {mode} member {m}\n"); sys.stderr.flush()
        for _ in range(3):
            K.cg_solve(desc, rd, precond=pre, tolerance=1e-4)
        torch.cuda.synchronize()
    os.environ.pop("LO_OC_DEBUG", None)
