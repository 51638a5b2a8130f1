"""Column-lockstep resident CG (lo_cg_lockstep.hip) on the cfg3 shape: 512 x 8192 (R = 32), 16 probe columns with
tridiagonals, 21 iterations.  LO_LS_DEBUG=<member> prints the in-kernel phase timers (100 MHz ticks)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = int(os.environ.get("LS_B", 512)), int(os.environ.get("LS_N", 8192)), 32
c = int(os.environ.get("LS_C", 16))
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
full = torch.randn(B, N, c, generator=g, device=dev); full /= full.norm(dim=-2, keepdim=True)
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)  # Q form + root form
if os.environ.get("LS_NOPRE"): pre = None  # (z = r: the kernel without Q, H, G)
nt = min(c, 16)
def run():
    return K.cg_solve(desc, full, precond=pre, n_tridiag=nt, tolerance=1e-4)
run(); torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
for _ in range(reps): r = run()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
print(f"B={B} N={N} c={c}: cg_solve {t*1e3:.3f} ms, iterations {r.iterations}")
_hip.prof_enable(True); run(); run(); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
for k, (cnt, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:6]: print(f"    {k:20s} {cnt:5d} x {ms / cnt * 1e3:9.1f} us")
flop = 2.0 * B * N * (2 * 32 + 2 * 16) * 16 * 21
if "cg_lockstep" in p:
    us = p["cg_lockstep"][1] / p["cg_lockstep"][0] * 1e3
    print(f"    lockstep: {flop / us / 1e6:.1f} TFLOP/s fp32 of 157.3 ({flop / us / 1e6 / 157.3:.2f})")
