"""Resident pivoted Cholesky (rank 15) over the member size."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for R in (32, 16):
    for N in (1024, 2048, 4096, 8192, 16384):
        B = 1024 if N <= 8192 else 512
        Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
        desc = K.lowrank_diag_descriptor(Cm, None)
        for _ in range(2): K.pivoted_cholesky(desc, 15, contiguous=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): K.pivoted_cholesky(desc, 15, contiguous=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"R={R:2d} N={N:6d} B={B:4d}: {dt*1e3:7.3f} ms per factorisation of the batch")
