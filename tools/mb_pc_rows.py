"""Pivoted Cholesky (rank 15) of dense / Kronecker operators: the resident factorisation k_pc_onchip_rows against the
streaming engine (LO_PC_NO_RESIDENT_ROWS=1); pivots and factor must be bit-identical."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)

def run(desc, rank):
    out = {}
    for mode in ("resident", "streaming"):
        os.environ.pop("LO_PC_NO_RESIDENT_ROWS", None)
        if mode == "streaming": os.environ["LO_PC_NO_RESIDENT_ROWS"] = "1"
        for _ in range(3): L, perm = K.pivoted_cholesky(desc, rank, contiguous=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): L, perm = K.pivoted_cholesky(desc, rank, contiguous=False)
        torch.cuda.synchronize(); out[mode] = ((time.perf_counter() - t0) / 10, L.clone(), perm.clone())
    os.environ.pop("LO_PC_NO_RESIDENT_ROWS", None)
    same = torch.equal(out["resident"][1], out["streaming"][1]) and torch.equal(out["resident"][2], out["streaming"][2])
    return out["resident"][0], out["streaming"][0], same

for (B, N) in ((1, 1000), (1, 4000), (1, 10001), (4, 16384), (32, 4096), (200, 700)):
    X = torch.randn(B, N, 64, generator=g, device=dev) / 8
    Kd = X @ X.mT + 0.05 * torch.eye(N, device=dev)
    tr, ts, same = run(K.dense_diag_descriptor(Kd, None), 15)
    print(f"dense B={B:4d} N={N:6d}: resident {tr*1e6:8.1f} us   streaming {ts*1e6:8.1f} us   bit-identical {same}", flush=True)
    del Kd
for (B, n1, n2) in ((4, 64, 64), (16, 128, 100), (128, 256, 256)):
    X1 = torch.randn(B, n1, n1, generator=g, device=dev) / n1 ** 0.5; X2 = torch.randn(B, n2, n2, generator=g, device=dev) / n2 ** 0.5
    K1 = X1 @ X1.mT + 0.1 * torch.eye(n1, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n2, device=dev)
    desc = K.kron_diag_descriptor(K1, K2, None)
    tr, ts, same = run(desc, 15)
    print(f"kron  B={B:4d} {n1} x {n2}: resident {tr*1e6:8.1f} us   streaming {ts*1e6:8.1f} us   bit-identical {same}", flush=True)
