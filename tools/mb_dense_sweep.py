"""Dense matvec over the member size (total K bytes held at ~2 GiB): rate per engine, c = 1 and 17."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for N in (256, 512, 1000, 1024, 2048, 4096, 8192, 16384):
    B = max(2, int(2 ** 31 / (N * N * 4)))
    Kd = torch.randn(B, N, N, generator=g, device=dev) / N ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    desc = K.dense_diag_descriptor(Kd, d)
    for c in (1, 3, 17):
        v = torch.randn(B, N, c, generator=g, device=dev)
        for _ in range(2): K.matvec(desc, v)
        torch.cuda.synchronize(); _hip.prof_enable(True)
        for _ in range(5): K.matvec(desc, v)
        torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
        tot = sum(ms for _, ms in p.values()) / 5
        print(f"N={N:6d} B={B:5d} c={c:2d}: {tot*1e3:8.1f} us = {B * N * N * 4 / tot / 1e9:5.2f} TB/s  ({', '.join(p)})")
    del Kd
