"""Explicit Lanczos at the survey's probe-run shape and at the full cfg3 batch."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
N, R, P, k = 8192, 32, 16, 20
for B in (4, 64, 512):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    d = torch.rand(B, N, generator=g, device="cuda") + 0.5
    V = torch.randn(B, N, P, generator=g, device="cuda")
    desc = K.lowrank_diag_descriptor(Cm, d)
    K.lanczos_tridiag(desc, V, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): q, t = K.lanczos_tridiag(desc, V, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"B={B}: lanczos_tridiag k={k} P={P}: {dt*1e3:.2f} ms, q {tuple(q.shape)} t {tuple(t.shape)}")
    if B == 512:
        _hip.prof_enable(True); K.lanczos_tridiag(desc, V, k); torch.cuda.synchronize()
        for kk, (c, ms) in sorted(_hip.prof_report().items()): print(f"    {kk:20s} {c:5d} x {ms / c * 1e3:9.1f} us  total {ms:8.2f} ms")
