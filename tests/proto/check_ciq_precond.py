"""Preconditioned contour integral quadrature (N above settings.min_preconditioning_size): ||A^-1/2 b||^2 against b^T A^-1 b."""
import sys, torch, numpy as np, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")): sys.path.insert(0, p_)
import cases
from linear_operator_amd import settings, _hip
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
C, d, rhs = cases.lowrank_diag(1601, 3, 4096, 32, 2)
dev = lambda a: torch.from_numpy(a).cuda()
A = AddedDiagLinearOperator(LowRankRootLinearOperator(dev(C)), DiagLinearOperator(dev(d)))
x = A.sqrt_inv_matmul(dev(rhs)); torch.cuda.synchronize()  # first call: scipy / rocSOLVER start-up
_hip.prof_enable(True)
t0 = time.perf_counter(); x = A.sqrt_inv_matmul(dev(rhs)); torch.cuda.synchronize(); print("ms", (time.perf_counter() - t0) * 1e3)
p = _hip.prof_report(); _hip.prof_enable(False)
print(sorted(p.items(), key=lambda kv: -kv[1][1])[:8])
from oracle import lo_oracle as orc
exact = orc.woodbury_solve(C.astype(np.float64), d.astype(np.float64), rhs.astype(np.float64))
iq_exact = (rhs.astype(np.float64) * exact).sum(-2)
iq = (x.double() ** 2).sum(-2).cpu().numpy()
print("||A^-1/2 b||^2 vs b^T A^-1 b:", np.abs(iq - iq_exact).max() / np.abs(iq_exact).max())
with settings.min_preconditioning_size(10 ** 9):
    x0 = A.sqrt_inv_matmul(dev(rhs))
print("unpreconditioned vs preconditioned root (differ by an orthogonal factor in general):", ((x - x0).norm() / x0.norm()).item())
iq0 = (x0.double() ** 2).sum(-2).cpu().numpy(); print(np.abs(iq0 - iq_exact).max() / np.abs(iq_exact).max())
