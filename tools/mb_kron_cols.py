"""Kronecker matvec (128 members, 256 x 256 factors, N = 65536) against the number of columns."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, n = 128, 256
K1 = torch.randn(B, n, n, generator=g, device=dev) / 16; K2 = torch.randn(B, n, n, generator=g, device=dev) / 16
sig = torch.full((B,), 1e-2, device=dev)
desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
for c in (1, 2, 4, 8, 17):
    v = torch.randn(B, n * n, c, generator=g, device=dev)
    for _ in range(2): K.matvec(desc, v)
    torch.cuda.synchronize(); _hip.prof_enable(True)
    for _ in range(3): y = K.matvec(desc, v)
    torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    tot = sum(ms for _, ms in p.values()) / 3
    flop = 2.0 * B * c * 2 * n ** 3
    print(f"c={c:2d}: {tot*1e3:8.1f} us per matvec = {flop / tot / 1e9:6.1f} TFLOP/s  [{', '.join(f'{k} {ms/cn*1e3:.0f}' for k,(cn,ms) in p.items())}]")
