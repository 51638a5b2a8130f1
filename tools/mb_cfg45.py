import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda")

def timeit(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r

def prof(fn):
    _hip.prof_enable(True); fn(); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    for k, (c, ms) in sorted(p.items()): print(f"    {k:20s} {c:5d} x {ms / c * 1e3:9.1f} us  total {ms:8.2f} ms")

which = sys.argv[1] if len(sys.argv) > 1 else "both"
g = torch.Generator(device=dev); g.manual_seed(3)
if which in ("cfg4", "both"):
    B, n = 128, 256
    X1 = torch.randn(B, n, n, generator=g, device=dev) / 16; X2 = torch.randn(B, n, n, generator=g, device=dev) / 16
    K1 = X1 @ X1.mT + 0.1 * torch.eye(n, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n, device=dev)
    sig = torch.full((B,), 1e-2, device=dev); rhs = torch.randn(B, n * n, 1, generator=g, device=dev)
    desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
    t, (L, perm) = timeit(lambda: K.pivoted_cholesky(desc.without_diag(), 15), 1); print(f"cfg4/GPU shard B={B}: pivoted cholesky {t*1e3:.2f} ms, rank {L.shape[-1]}")
    pre = K.precond_build(L, sig, True, perm=perm, kron=desc)  # (+ the Kronecker root form)
    t, res = timeit(lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-3), 1)
    print(f"  CG: {t*1e3:.1f} ms, iterations {res.iterations}, {B*res.matvecs/t/1e3:.1f} k member-matvecs/s, mean resid {res.mean_residual:.2e}")
    prof(lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-3))
if which in ("cfg5", "both"):
    B, N, c = 8, 16384, 17
    X = torch.randn(B, N, N, generator=g, device=dev) / 128
    Kd = X @ X.mT; del X
    d = torch.rand(B, N, generator=g, device=dev) + 0.5; rhs = torch.randn(B, N, c, generator=g, device=dev)
    desc = K.dense_diag_descriptor(Kd, d)
    t, (L, _) = timeit(lambda: K.pivoted_cholesky(desc, 15), 1); print(f"cfg5 B={B} (of 32 per GPU): pivoted cholesky {t*1e3:.2f} ms")
    pre = K.precond_build(L, d, False)
    t, res = timeit(lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4, n_tridiag=16), 1)
    print(f"  CG: {t*1e3:.1f} ms, iterations {res.iterations}, {B*res.matvecs/t:.1f} member-matvecs/s (17 cols)")
    prof(lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4, n_tridiag=16))
