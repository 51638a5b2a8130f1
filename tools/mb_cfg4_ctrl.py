"""cfg4 shard CG wall time with / without the control step folded into the fused preconditioner apply."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
B, n = 128, 256
X1 = torch.randn(B, n, n, generator=g, device=dev) / 16; X2 = torch.randn(B, n, n, generator=g, device=dev) / 16
K1 = X1 @ X1.mT + 0.1 * torch.eye(n, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n, device=dev)
sig = torch.full((B,), 1e-2, device=dev); rhs = torch.randn(B, n * n, 1, generator=g, device=dev)
desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
L, _ = K.pivoted_cholesky(desc, 15)
pre = K.precond_build(L, sig, True)
def run(): return K.cg_solve(desc, rhs, precond=pre, tolerance=1e-3)
for rep in range(3):
    for flag in ("", "1"):
        if flag: os.environ["LO_NO_FUSED_CTRL"] = "1"
        else: os.environ.pop("LO_NO_FUSED_CTRL", None)
        run(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): r = run()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
        print(f"fused ctrl {'off' if flag else 'on '}: {t*1e3:.2f} ms, iterations {r.iterations}, resid {r.mean_residual:.3e}")
