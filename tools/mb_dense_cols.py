"""Dense matvec (8 members of 16384^2) against the number of columns: which engine streams K fastest?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N = 8, 16384
Kd = torch.randn(B, N, N, generator=g, device=dev) / 128
d = torch.rand(B, N, generator=g, device=dev) + 0.5
desc = K.dense_diag_descriptor(Kd, d)
for c in (1, 2, 3, 4, 5, 8, 16, 17, 20, 24, 32):
    v = torch.randn(B, N, c, generator=g, device=dev)
    for _ in range(2): K.matvec(desc, v)
    torch.cuda.synchronize(); _hip.prof_enable(True)
    for _ in range(5): y = K.matvec(desc, v)
    torch.cuda.synchronize(); p = _hip.prof_report(); _hip.prof_enable(False)
    tot = sum(ms for _, ms in p.values()) / 5
    ref = Kd[0].double() @ v[0].double() + d[0].double().unsqueeze(-1) * v[0].double()
    err = ((y[0].double() - ref).norm() / ref.norm()).item()
    print(f"c={c:2d}: {tot*1e3:8.1f} us per matvec = {B * N * N * 4 / tot / 1e9:6.2f} TB/s   ({', '.join(p.keys())})  err {err:.1e}")
