"""float64 operators through the preconditioned path (round 4): inv_quad_logdet forward + backward against dense float64
autograd, and what is still fp32-only (explicit Lanczos)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases, linear_operator_amd as lo
from linear_operator_amd.operators import *
C, d, rhs = cases.lowrank_diag(2501, 2, 2048, 16, 1, dtype=np.float64)
Cg = torch.from_numpy(C).cuda().requires_grad_(True); dg = torch.from_numpy(d).cuda().requires_grad_(True)
A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
torch.manual_seed(0)
with lo.settings.cg_tolerance(1e-8), lo.settings.num_trace_samples(64):
    iq, ld = A.inv_quad_logdet(torch.from_numpy(rhs).cuda(), logdet=True)
    (iq.sum() + ld.sum()).backward()
dense = (Cg.detach() @ Cg.detach().mT + torch.diag_embed(dg.detach())).requires_grad_(True)
r = torch.from_numpy(rhs).cuda()
(( r * torch.linalg.solve(dense, r)).sum() + torch.logdet(dense).sum()).backward()
gd = dense.grad.diagonal(dim1=-1, dim2=-2)
print("logdet", ld.tolist(), torch.logdet(dense).tolist())
print("diag grad rel err", float((dg.grad - gd).norm() / gd.norm()))
# lanczos fp64
try:
    print(LowRankRootLinearOperator(Cg.detach()).add_jitter(1.0).root_decomposition().root.shape)
except Exception as e:
    print("root_decomposition fp64:", type(e).__name__, str(e)[:120])
