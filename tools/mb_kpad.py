"""cfg4 operator through the reference's DEFAULT routing (Kronecker + constant diagonal -> eigendecomposition closed
form) against the CG path of the explicit AddedDiagLinearOperator, 128 members of 256 (x) 256."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings
from linear_operator_amd.operators import (AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator,
                                           KroneckerProductLinearOperator)
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(5)
B, n = 128, 256
X1 = torch.randn(B, n, n, generator=g, device=dev) / 16; X2 = torch.randn(B, n, n, generator=g, device=dev) / 16
K1 = X1 @ X1.mT + 0.1 * torch.eye(n, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n, device=dev)
sig = torch.full((B, 1), 1e-2, device=dev); rhs = torch.randn(B, n * n, 1, generator=g, device=dev)
kp = KroneckerProductLinearOperator(DenseLinearOperator(K1), DenseLinearOperator(K2))
def t(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
with settings.cg_tolerance(1e-3):
    ms_e, xe = t(lambda: (kp + ConstantDiagLinearOperator(sig, n * n)).solve(rhs))
    ms_c, xc = t(lambda: AddedDiagLinearOperator(kp, ConstantDiagLinearOperator(sig, n * n)).solve(rhs))
    ms_l, ld = t(lambda: (kp + ConstantDiagLinearOperator(sig, n * n)).logdet())
t0 = time.perf_counter(); ev = torch.linalg.eigh(K1.double()); torch.cuda.synchronize(); ms_eig = (time.perf_counter() - t0) * 1e3
print(f"eig closed form solve {ms_e:.1f} ms | CG solve {ms_c:.1f} ms | logdet (eig) {ms_l:.1f} ms | one fp64 batched eigh {ms_eig:.1f} ms")
print("rel diff eig vs CG:", ((xe - xc).norm() / xe.norm()).item())
