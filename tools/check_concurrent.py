"""The resident kernels need all their workgroups co-resident.  Here another stream keeps the GPU busy with unrelated
kernels (as RCCL's all-gather does in the multi-GPU bench) while solves run: results must be bit-identical to the quiet
run and no group exchange may time out (a timeout prints a fallback message and is counted)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
rhs = torch.randn(B, N, 1, generator=g, device=dev)
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)  # Q form + root form
ref = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4).x.clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)
big = torch.randn(256 * 1024 * 1024 // 4, device=dev)
def load(kind, n):
    with torch.cuda.stream(side):
        for _ in range(n):
            if kind == "gemm": (a @ b)
            else: big.mul_(1.0000001)
for kind in ("gemm", "stream"):
    load(kind, 200)
    t0 = time.perf_counter(); bad = 0
    for i in range(50):
        x = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4).x
        Lx, _ = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
        bad += int(not torch.equal(x, ref)) + int(not torch.equal(Lx, L))
    torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
    print(f"background {kind}: 50 x (solve + pivoted Cholesky) in {dt*1e3:.1f} ms, mismatches {bad}")
    torch.cuda.synchronize()
