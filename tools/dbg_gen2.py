import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
B, N, R = int(os.environ.get("B", 70)), int(os.environ.get("N", 8192)), 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d)
L, _ = K.pivoted_cholesky(desc, 15, contiguous=False); pre = K.precond_build(L, d, False)
res = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
x = res.x
print("iters", res.iterations, "nan members", torch.isnan(x).flatten(1).any(1).nonzero().flatten().tolist()[:20])
bad = torch.isnan(x[:, :, 0])
if bad.any():
    b = int(bad.any(1).nonzero()[0])
    rows = bad[b].nonzero().flatten()
    print("member", b, "nan rows", rows.numel(), rows[:10].tolist(), rows[-5:].tolist())
os.environ["LO_OC_GEN1"] = "1"
ref = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
err = ((x - ref.x).flatten(1).norm(dim=1) / ref.x.flatten(1).norm(dim=1))
print("gen1 iters", ref.iterations, "rel err per member (first 12)", [f"{e:.1e}" for e in err[:12].tolist()], "worst", float(err.max()), int(err.argmax()))
