"""cfg3 inv_quad_logdet at the C-ABI level (what bench.py's cfg3 extra times): per-kernel breakdown."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
if os.environ.get("LO_LIB_VARIANT"):  # experiments: variants/liblo_amd_<name>.so (tools/build_variant.sh)
    _hip._LIB_PATH = os.path.join(ROOT, "variants", "liblo_amd_%s.so" % os.environ["LO_LIB_VARIANT"])
import bench
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(77)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
full = torch.randn(B, N, 17, generator=g, device=dev); full[..., :16] /= full[..., :16].norm(dim=-2, keepdim=True)
desc = K.lowrank_diag_descriptor(Cm, d)
def iql():
    pre = bench.build_precond(desc, d, need_q=False)
    r = K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=1e-4)
    _, _, ld = K.tridiag_eigh_slq(r.t_mat, N)
    return r, ld + pre.logdet
for _ in range(3): iql()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); iql(); torch.cuda.synchronize(); print(f"iql {1e3*(time.perf_counter()-t0):.2f} ms")
def timed(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
pre = bench.build_precond(desc, d)
print(f"build_precond {timed(lambda: bench.build_precond(desc, d)):.2f} ms; cg {timed(lambda: K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=1e-4)):.2f} ms")
r = K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=1e-4)
print(f"eig+slq {timed(lambda: K.tridiag_eigh_slq(r.t_mat, N)):.2f} ms")
_hip.prof_enable(True); iql(); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1]): print(f"    {k:24s} {c:5d} x {ms / c * 1e3:9.1f} us  total {ms:8.2f} ms")
