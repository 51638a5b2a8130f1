"""t_mat of the R-space columns path and of the lockstep / serial resident kernels against the fp64 oracle iteration."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from linear_operator_amd import kernels as K
from oracle import lo_oracle as orc
dev = torch.device("cuda")
N, R, B, P = 4096, 32, 3, 4
C, d, rhs = cases.lowrank_diag(7901, B, N, R, P + 1)
rhs[..., :P] /= np.linalg.norm(rhs[..., :P], axis=-2, keepdims=True)
Cd, dd, rd = (torch.from_numpy(a).to(dev) for a in (C, d, rhs))
desc = K.lowrank_diag_descriptor(Cd, dd)
L, perm = K.pivoted_cholesky(desc, 15)
pre = K.precond_build(L, dd, constant_diag=False, root=desc.A0, perm=perm)
Lh = L.cpu().numpy()
C64, d64 = C.astype(np.float64), d.astype(np.float64)
pre64 = orc.Preconditioner(Lh.astype(np.float64), d64)
x64, t64, info = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C64, d64, v), rhs.astype(np.float64), n_tridiag=P, tolerance=1e-4,
                               preconditioner=pre64.apply)
pre32 = orc.Preconditioner(Lh, d)
x32, t32, info32 = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, n_tridiag=P, tolerance=1e-4, preconditioner=pre32.apply)
print("oracle64 t", t64.shape, "oracle32 t", t32.shape, "iters", info.iterations, info32.iterations)
for name, env in (("rs", {}), ("old", {"LO_NO_RSPACE_COLS": "1"})):
    os.environ.pop("LO_NO_RSPACE_COLS", None); os.environ.update(env)
    res = K.cg_solve(desc, rd, precond=pre, tolerance=1e-4, n_tridiag=P)
    t = res.t_mat.cpu().numpy().astype(np.float64)
    m = min(t.shape[-1], t64.shape[-1], 6)
    print(name, K.cg_last_executed()["rspace"], "t", t.shape, "lead block max rel diff vs oracle64:",
          np.abs(t[..., :m, :m] - t64[..., :m, :m]).max() / np.abs(t64[..., :m, :m]).max(),
          "vs oracle32:", np.abs(t[..., :m, :m] - t32[..., :m, :m]).max() / np.abs(t32[..., :m, :m]).max())
    print("   diag", t[0, 0].diagonal()[:6], "\n   or64", t64[0, 0].diagonal()[:6], "\n   or32", t32[0, 0].diagonal()[:6])
    _, _, pk = K.tridiag_eigh_slq(res.t_mat, N)
    ev, evec = orc.lanczos_tridiag_to_diag(t64)
    print("   slq", pk.cpu().numpy(), "oracle64 slq", orc.slq_logdet(N, ev, evec))
