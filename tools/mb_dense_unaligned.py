"""Dense matvec for rows that are not 16-byte aligned (N % 4 != 0): correctness against float64 over N mod 4, batch
offsets, split-K and ragged edges, and the rate next to an aligned N of the same size."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
bad = 0
for (B, N) in ((1, 257), (3, 513), (2, 1001), (3, 1002), (2, 1003), (1, 4001), (5, 2050), (2, 300), (1, 4099)):
    for c in (2, 11, 17):
        Kd = torch.randn(B, N, N, generator=g, device=dev) / N ** 0.5
        d = torch.rand(B, N, generator=g, device=dev) + 0.5
        v = torch.randn(B, N, c, generator=g, device=dev)
        y = K.matvec(K.dense_diag_descriptor(Kd, d), v)
        ref = Kd.double() @ v.double() + d.double().unsqueeze(-1) * v.double()
        err = ((y.double() - ref).norm() / ref.norm()).item()
        # a member selected out of a bigger tensor: the base pointer itself is misaligned
        big = torch.randn(B + 1, N, N, generator=g, device=dev) / N ** 0.5
        y2 = K.matvec(K.dense_diag_descriptor(big[1:], d), v)
        ref2 = big[1:].double() @ v.double() + d.double().unsqueeze(-1) * v.double()
        err2 = ((y2.double() - ref2).norm() / ref2.norm()).item()
        ok = err < 2e-6 and err2 < 2e-6
        bad += not ok
        print(f"B={B} N={N} c={c}: rel err {err:.1e} / {err2:.1e} {'ok' if ok else 'MISMATCH'}")
for (B, N) in ((7, 10000), (7, 10001), (7, 10002), (7, 10003), (3, 9999), (2, 16383), (1, 4001)):
    Kd = torch.randn(B, N, N, generator=g, device=dev) / 128
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    desc = K.dense_diag_descriptor(Kd, d)
    for c in (11, 17):
        v = torch.randn(B, N, c, generator=g, device=dev)
        best = 1e9
        for rnd in range(3):
            for _ in range(2): K.matvec(desc, v)
            torch.cuda.synchronize(); _hip.prof_enable(True)
            for _ in range(5): K.matvec(desc, v)
            torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
            best = min(best, sum(ms for _, ms in p.values()) / 5)
        print(f"B={B} N={N} c={c}: {best*1e3:8.1f} us = {B*N*N*4/best/1e9:5.2f} TB/s")
    del Kd, desc
print("OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
