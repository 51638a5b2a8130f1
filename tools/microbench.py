#!/usr/bin/env python3
"""Per-kernel HIP-event timing of the CG solve for a sweep of batch sizes / columns (development aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linear_operator_amd import _hip  # noqa: E402
from linear_operator_amd import kernels as K  # noqa: E402


def run(B, N, R, c, k=15, reps=3, tol=1e-4, n_tridiag=0):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    rhs = torch.randn(B, N, c, generator=g, device=dev)
    desc = K.lowrank_diag_descriptor(Cm, d)
    L, _ = K.pivoted_cholesky(desc, k)
    pre = K.precond_build(L, d, False)
    for _ in range(2):
        res = K.cg_solve(desc, rhs, precond=pre, tolerance=tol, n_tridiag=n_tridiag)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        res = K.cg_solve(desc, rhs, precond=pre, tolerance=tol, n_tridiag=n_tridiag)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    _hip.prof_enable(True)
    for _ in range(reps):
        K.cg_solve(desc, rhs, precond=pre, tolerance=tol, n_tridiag=n_tridiag)
    torch.cuda.synchronize()
    prof = _hip.prof_report()
    _hip.prof_enable(False)
    print(f"B={B} N={N} R={R} c={c} iters={res.iterations}: {ms:.3f} ms/solve  "
          f"({B * res.matvecs / ms * 1e3 / 1e6:.3f} M member-matvecs/s)")
    for name, (cnt, tot) in sorted(prof.items()):
        avg = tot / cnt * 1e3
        if name.startswith("skinny"):
            r = R if name.endswith(f"R{max(4, 1 << (max(R, 4) - 1).bit_length())}") else k
            nb = 4 * B * (N * r + (N if "nn" in name else 0) + (2 if "nn" in name else 1) * N * c)
            print(f"    {name:18s} {cnt:4d} x {avg:8.1f} us   {nb / avg / 1e3:8.1f} GB/s (algorithmic)")
        else:
            print(f"    {name:18s} {cnt:4d} x {avg:8.1f} us")


if __name__ == "__main__":
    cfgs = [(64, 8192, 32, 1), (128, 8192, 32, 1), (256, 8192, 32, 1), (512, 8192, 32, 1)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for cfg in cfgs:
        run(*cfg)
