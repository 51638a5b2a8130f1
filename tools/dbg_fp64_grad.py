import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden"); sys.path.insert(0, "/root/repo/tests")
import cases
import linear_operator_amd as lo
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
clear_preconditioner_memo()
C, d, rhs = cases.lowrank_diag(2601, 2, 2048, 16, 1, dtype=np.float64)
Cg, dg = dev(C).requires_grad_(True), dev(d).requires_grad_(True)
y = dev(rhs)
dense = (Cg.detach() @ Cg.detach().mT + torch.diag_embed(dg.detach())).requires_grad_(True)
(torch.linalg.solve(dense, y).mul(y).sum() + torch.logdet(dense).sum()).backward()
want_d = dense.grad.diagonal(dim1=-1, dim2=-2)
want_C = 2 * ((dense.grad + dense.grad.mT) / 2) @ Cg.detach()
with lo.settings.cg_tolerance(1e-10), lo.settings.num_trace_samples(64), lo.settings.max_cg_iterations(200):
    A0 = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
    with torch.no_grad():
        A0.solve(y)
    for rep in range(2):
        torch.manual_seed(11 + rep)
        Cg.grad = dg.grad = None
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
        iq, ld = A.inv_quad_logdet(y, logdet=True)
        (iq.sum() + ld.sum()).backward()
        x = torch.linalg.solve(dense.detach(), y)
        print(rep, float((dg.grad - want_d).norm() / want_d.norm()), float((Cg.grad - want_C).norm() / want_C.norm()),
              float((dg.grad + x.squeeze(-1) ** 2).norm()), float((dg.grad - want_d).norm()), float(ld.sum()), float(torch.logdet(dense).sum()))
