"""Few very tall low-rank + diagonal members (streaming engine): does the row split fill the chip?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for B, N in ((1, 262144), (1, 1048576), (4, 1048576), (16, 262144)):
    R = 32
    Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    rhs = torch.randn(B, N, 1, generator=g, device=dev)
    desc = K.lowrank_diag_descriptor(Cm, d)
    L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
    pre = K.precond_build(L, d, False)
    for _ in range(2): r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    byts = 4.0 * B * N * (2 * R + 2 * 16 + 12) * r.iterations
    _hip.prof_enable(True); K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4); torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    print(f"B={B} N={N}: CG {dt*1e3:.2f} ms, {r.iterations} it, streamed-bytes rate {byts / dt / 1e12:.2f} TB/s  [" +
          ", ".join(f"{k} {ms/c*1e3:.0f}us" for k, (c, ms) in sorted(p.items(), key=lambda kv: -kv[1][1])[:4]) + "]")
