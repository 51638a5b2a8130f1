import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from linear_operator_amd import settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DenseLinearOperator, DiagLinearOperator
for N in (120, 500):
  for seed in range(3):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, 48).astype(np.float32) / 48 ** 0.5
    d = (rs.rand(N) + 0.3).astype(np.float32)
    rhs = rs.randn(N, 2).astype(np.float32)
    Kd = (torch.from_numpy(X) @ torch.from_numpy(X).T).cuda()
    A64 = Kd.double() + torch.diag(torch.from_numpy(d).cuda().double())
    ev, V = torch.linalg.eigh(A64)
    ex = V @ torch.diag(ev.rsqrt()) @ V.T @ torch.from_numpy(rhs).cuda().double()
    for chol, minp, tol in ((0, 2000, 1e-3), (0, 100, 1e-3), (800, 2000, 1e-3), (0, 2000, 1.0)):
        A = AddedDiagLinearOperator(DenseLinearOperator(Kd), DiagLinearOperator(torch.from_numpy(d).cuda()))
        with settings.max_cholesky_size(chol), settings.min_preconditioning_size(minp), settings.cg_tolerance(tol):
            out = A.sqrt_inv_matmul(torch.from_numpy(rhs).cuda())
        print(N, seed, (chol, minp, tol), "rel err", float((out.double() - ex).norm() / ex.norm()), flush=True)
