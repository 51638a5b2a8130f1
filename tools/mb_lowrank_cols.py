"""Low-rank + diagonal matvec (512 members, N = 8192, R = 32) against the number of columns: VALU / MFMA skinny engines."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N = 512, 8192
for R in (32, 16):
    Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    desc = K.lowrank_diag_descriptor(Cm, d)
    for c in (1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 24, 32):
        v = torch.randn(B, N, c, generator=g, device=dev)
        for _ in range(2): K.matvec(desc, v)
        torch.cuda.synchronize(); _hip.prof_enable(True)
        for _ in range(5): y = K.matvec(desc, v)
        torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
        tot = sum(ms for _, ms in p.values()) / 5
        byts = 4.0 * B * N * (2 * R + 1 + 3 * c)  # C twice (two passes), d, v in twice, y out
        print(f"R={R} c={c:2d}: {tot*1e3:7.1f} us = {byts / tot / 1e9:5.2f} TB/s (C streamed twice)  [{', '.join(f'{k} {ms/cn*1e3:.0f}' for k,(cn,ms) in p.items())}]")
