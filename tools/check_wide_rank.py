"""Low-rank matvec / CG with root ranks above 32 (skinny kernels with 16, 32, 64 quads per row) against fp64."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
for R in (33, 40, 64, 100, 128, 200, 256):
    for c in (1, 3, 8, 17):
        B, N = 3, 1500
        Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
        d = torch.rand(B, N, generator=g, device=dev) + 0.5
        v = torch.randn(B, N, c, generator=g, device=dev)
        y = K.matvec(K.lowrank_diag_descriptor(Cm, d), v)
        ref = Cm.double() @ (Cm.double().mT @ v.double()) + d.double().unsqueeze(-1) * v.double()
        err = ((y.double() - ref).norm(dim=-2) / ref.norm(dim=-2)).max().item()
        print(f"R={R} c={c}: matvec err {err:.2e}" + ("  <-- BAD" if err > 1e-5 else ""))
