"""Dense matvec: three workgroups per CU against the two-per-CU cap the launcher picks when it removes a thin last round
(LO_DENSE_OCC3=1 disables the cap).  Interleaved, best of four."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for (B, N) in ((4, 16384), (8, 16384), (12, 16384), (16, 16384), (7, 10000), (5, 12288), (13, 8192), (20, 4096)):
    Kd = torch.randn(B, N, N, generator=g, device=dev) / 128
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    desc = K.dense_diag_descriptor(Kd, d)
    for c in (11, 17):
        v = torch.randn(B, N, c, generator=g, device=dev)
        out = {}
        for rnd in range(4):
            for mode in ("auto", "occ3"):
                os.environ.pop("LO_DENSE_OCC3", None)
                if mode == "occ3": os.environ["LO_DENSE_OCC3"] = "1"
                for _ in range(2): K.matvec(desc, v)
                torch.cuda.synchronize(); _hip.prof_enable(True)
                for _ in range(5): K.matvec(desc, v)
                torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
                tt = sum(ms for _, ms in p.values()) / 5
                out[mode] = min(out.get(mode, 1e9), tt)
        os.environ.pop("LO_DENSE_OCC3", None)
        W = B * ((N + 63) // 64)
        print(f"B={B:3d} N={N:6d} c={c:2d} ({W} workgroups): auto {out['auto']*1e3:8.1f} us = {B*N*N*4/out['auto']/1e9:5.2f} TB/s   "
              f"three per CU {out['occ3']*1e3:8.1f} us = {B*N*N*4/out['occ3']/1e9:5.2f} TB/s", flush=True)
    del Kd, desc
