"""Rate of the Kronecker matvec's matrix-core GEMMs (k_kron_nt_mfma) against the contraction length: the BASELINE cfg4
factors (256 x 256) give K = 256 = four 64-wide slabs per 128 x 128 tile, so the tile's prologue (first global fetch, not
overlapped with anything of the same workgroup) and epilogue (64 strided stores per lane + the fused dot) weigh as much
as the slab loop.  Larger factors amortise them; the slab loop itself runs near the bare fp32 MFMA rate."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(5)
for n, B in ((128, 512), (256, 128), (256, 512), (512, 32), (512, 128), (1024, 8), (1024, 32)):
    K1 = torch.randn(B, n, n, generator=g, device=dev) / n ** 0.5
    K2 = torch.randn(B, n, n, generator=g, device=dev) / n ** 0.5
    d = torch.rand(B, n * n, generator=g, device=dev) + 0.5
    v = torch.randn(B, n * n, 1, generator=g, device=dev)
    desc = K.kron_diag_descriptor(K1, K2, d)
    for _ in range(3): K.matvec(desc, v)
    torch.cuda.synchronize()
    _hip.prof_enable(True)
    for _ in range(10): K.matvec(desc, v)
    torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    cnt, ms = p["kron_gemm_mfma"]
    us = ms / cnt * 1e3
    flop = 2.0 * B * n ** 3
    print(f"n1 = n2 = {n:5d} (K = {n}), B = {B:4d}: {cnt} GEMM launches, {us:8.1f} us each, {flop / us / 1e6:6.1f} TFLOP/s fp32, "
          f"{B * (n // 128) ** 2} tiles on 512 slots")
