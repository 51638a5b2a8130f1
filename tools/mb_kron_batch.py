"""Kronecker matvec / CG against the batch size (256 x 256 factors): how well do few members fill the chip?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for n in (256, 512):
    for B in (1, 2, 8, 32, 128):
        if n == 512 and B > 32: continue
        X1 = torch.randn(B, n, n, generator=g, device=dev) / n ** 0.5; X2 = torch.randn(B, n, n, generator=g, device=dev) / n ** 0.5
        K1 = X1 @ X1.mT + 0.1 * torch.eye(n, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n, device=dev)
        sig = torch.full((B,), 1e-2, device=dev); v = torch.randn(B, n * n, 1, generator=g, device=dev)
        desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
        for _ in range(2): K.matvec(desc, v)
        torch.cuda.synchronize(); _hip.prof_enable(True)
        for _ in range(5): K.matvec(desc, v)
        torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
        tot = sum(ms for _, ms in p.values()) / 5
        print(f"n={n} B={B:4d}: matvec {tot*1e3:7.1f} us = {2 * 2.0 * B * n**3 / tot / 1e9:6.1f} TFLOP/s")
