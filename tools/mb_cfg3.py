"""cfg3 pieces: CG with 17 columns + 16 tridiagonals on 512 x 8192 (R = 32), resident vs streaming."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
full = torch.randn(B, N, 17, generator=g, device=dev); full[..., :16] /= full[..., :16].norm(dim=-2, keepdim=True)
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)  # Q form + root form
def run():
    return K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=1e-4)
for on in (True, False):
    K.set_onchip_cg(on)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"resident={on}: CG 17 cols + tridiag: {t*1e3:.2f} ms, iterations {r.iterations}, T {tuple(r.t_mat.shape)}")
    _hip.prof_enable(True); run(); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:6]: print(f"    {k:20s} {c:5d} x {ms / c * 1e3:9.1f} us  total {ms:8.2f} ms")
K.set_onchip_cg(True)
t0 = time.perf_counter(); ev = K.tridiag_eigh_slq(r.t_mat, N); torch.cuda.synchronize(); print(f"tridiag eig + slq: {(time.perf_counter()-t0)*1e3:.2f} ms")
