"""Root-form preconditioner build (k_pb_gram_root + k_pb_rootform) against the batch size: latency or throughput?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
for B in (8, 64, 512, 2048):
    N, R = 8192, 32
    Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
    for _ in range(3): K.precond_build(L, d, False, root=Cm, perm=perm, need_q=False)
    torch.cuda.synchronize()
    _hip.prof_enable(True)
    for _ in range(5): K.precond_build(L, d, False, root=Cm, perm=perm, need_q=False)
    torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    print(f"B={B}: " + ", ".join(f"{k} {ms / c * 1e3:.1f} us" for k, (c, ms) in sorted(p.items())))
