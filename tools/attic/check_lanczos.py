"""Resident Lanczos (lo_lanczos_resident_f32) against the streaming engine on the same inputs, then timings."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
B, N, R, P, k = (int(os.environ.get(n, v)) for n, v in (("LZ_B", 64), ("LZ_N", 8192), ("LZ_R", 32), ("LZ_P", 16), ("LZ_K", 20)))
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
V = torch.randn(B, N, P, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d)
def run(resident):
    if resident: os.environ.pop("LO_NO_RESIDENT_LANCZOS", None)
    else: os.environ["LO_NO_RESIDENT_LANCZOS"] = "1"
    _hip.prof_enable(True)
    q, t = K.lanczos_tridiag(desc, V, k)
    torch.cuda.synchronize(); prof = _hip.prof_report(); _hip.prof_enable(False)
    return q, t, prof
qs, ts, ps = run(False)
qr, tr, pr = run(True)
print("kernels resident:", {n: round(v[1] / v[0] * 1e3, 1) for n, v in pr.items()})
print("shapes", tuple(qr.shape), tuple(tr.shape), tuple(qs.shape), tuple(ts.shape))
print("t max abs diff", (tr - ts).abs().max().item(), "of", ts.abs().max().item())
print("q max abs diff", (qr - qs).abs().max().item())
A = Cm @ Cm.mT + torch.diag_embed(d) if B * N * N * 4 < 2e9 else None
qd = qr[0, 0].double(); td = tr[0, 0].double()
print("orthonormality |Q^T Q - I|", (qd.mT @ qd - torch.eye(qd.shape[-1], device="cuda", dtype=torch.float64)).abs().max().item())
if A is not None:
    Ad = A[0].double()
    print("|Q^T A Q - T| (resident)", (qd.mT @ Ad @ qd - td).abs().max().item(), "(streaming)",
          (qs[0, 0].double().mT @ Ad @ qs[0, 0].double() - ts[0, 0].double()).abs().max().item())
for nm, res in (("streaming", False), ("resident", True)):
    if res: os.environ.pop("LO_NO_RESIDENT_LANCZOS", None)
    else: os.environ["LO_NO_RESIDENT_LANCZOS"] = "1"
    K.lanczos_tridiag(desc, V, k); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): K.lanczos_tridiag(desc, V, k)
    torch.cuda.synchronize(); print(nm, "ms", (time.perf_counter() - t0) / 3 * 1e3)
