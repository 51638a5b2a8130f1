"""Kronecker root form of the preconditioner (lo_precond_desc.kron_*) against the Q form: same solves, iteration counts,
time per iteration.  KR_B / KR_N1 / KR_N2 pick the shape (default: the cfg4 shard, 128 x (256 (x) 256))."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda")
B, n1, n2 = int(os.environ.get("KR_B", 128)), int(os.environ.get("KR_N1", 256)), int(os.environ.get("KR_N2", 256))
kind = os.environ.get("KR_KIND", "random")
g = torch.Generator(device=dev); g.manual_seed(3)
if kind == "random":
    X1 = torch.randn(B, n1, n1, generator=g, device=dev) / n1 ** 0.5; X2 = torch.randn(B, n2, n2, generator=g, device=dev) / n2 ** 0.5
    K1 = X1 @ X1.mT + 0.1 * torch.eye(n1, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n2, device=dev)
else:  # RBF kernels on random 1-d inputs: smooth, nearly dependent pivot rows
    ls = float(os.environ.get("KR_LS", 0.1))
    t1 = torch.rand(B, n1, 1, generator=g, device=dev); t2 = torch.rand(B, n2, 1, generator=g, device=dev)
    K1 = torch.exp(-0.5 * (t1 - t1.mT) ** 2 / ls ** 2); K2 = torch.exp(-0.5 * (t2 - t2.mT) ** 2 / ls ** 2)
sig = torch.full((B,), 1e-2, device=dev)
rhs = torch.randn(B, n1 * n2, 1, generator=g, device=dev)
desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
L, perm = K.pivoted_cholesky(desc.without_diag(), 15, contiguous=False)
pre_q = K.precond_build(L, sig, True)
K.KRON_ROOT_MAX_KAPPA = float(os.environ.get("KR_MAXK", K.KRON_ROOT_MAX_KAPPA))
pre_k = K.precond_build(L, sig, True, perm=perm, kron=desc)
lib = _hip.load()
print("kron root available:", pre_k.kron is not None, "worst kappa", pre_k.kron_kappa)
def run(pre):
    return K.cg_solve(desc, rhs, precond=pre, tolerance=float(os.environ.get("KR_TOL", 1e-3)))
out = {}
for name, pre in (("Q form", pre_q), ("kron root", pre_k)):
    run(pre); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = run(pre); torch.cuda.synchronize(); t = time.perf_counter() - t0
    out[name] = r
    print(f"{name:10s}: {t*1e3:8.2f} ms, iterations {r.iterations}, {t / max(r.iterations, 1) * 1e6:7.1f} us / iteration, "
          f"mean resid {r.mean_residual:.3e}")
    _hip.prof_enable(True); run(pre); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:5]: print(f"      {k:22s} {c:5d} x {ms / c * 1e3:8.1f} us")
a, b = out["Q form"].x, out["kron root"].x
print("rel diff of the solutions (per member max):", float(((a - b).norm(dim=-2) / a.norm(dim=-2)).max()))
# true residuals
for name in out:
    x = out[name].x
    res = (K.matvec(desc, x) - rhs).norm(dim=-2) / rhs.norm(dim=-2)
    print(f"  true relative residual {name}: max {float(res.max()):.3e}")
