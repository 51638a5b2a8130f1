"""Randomised Kronecker / dense operators through the streaming CG engine (matrix-core Kronecker matvec with and without
guards, split-K dense matvec, the fused preconditioner apply with the folded control step for N >= 8192) against fp64
dense solves.  Not part of the test suite: `python tools/fuzz_streaming.py --minutes 5 --seed 1` on the GPU box."""
import argparse, os, random, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rnd = random.Random(args.seed)
t_end = time.time() + 60 * args.minutes
dev = torch.device("cuda")
n_k = n_d = 0
while time.time() < t_end:
    g = torch.Generator(device=dev); g.manual_seed(rnd.randrange(1 << 30))
    c = rnd.choice([1, 1, 3, 11, 17, 32])
    k = rnd.choice([0, 7, 15, 32])
    if rnd.random() < 0.6:
        n1, n2 = rnd.choice([(64, 64), (72, 68), (96, 96), (128, 64), (100, 100), (33, 40), (64, 132), (200, 200)])
        B = rnd.choice([1, 2, 5]) if n1 * n2 <= 16384 else 1
        X1 = torch.randn(B, n1, n1, generator=g, device=dev) / n1 ** 0.5
        X2 = torch.randn(B, n2, n2, generator=g, device=dev) / n2 ** 0.5
        K1 = X1 @ X1.mT + 0.1 * torch.eye(n1, device=dev); K2 = X2 @ X2.mT + 0.1 * torch.eye(n2, device=dev)
        sig = torch.rand(B, generator=g, device=dev) * 0.1 + 0.02
        N = n1 * n2
        rhs = torch.randn(B, N, c, generator=g, device=dev)
        desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
        dense = torch.stack([torch.kron(K1[i].double(), K2[i].double()) for i in range(B)]) \
            + sig.double()[:, None, None] * torch.eye(N, device=dev, dtype=torch.float64)
        darg, const = sig, True
        n_k += 1
    else:
        N = rnd.choice([65, 300, 777, 1025, 1500, 2050, 4096, 5001, 8192, 20000])
        B = rnd.choice([1, 2, 3]) if N <= 4096 else 1
        X = torch.randn(B, N, N, generator=g, device=dev) / N ** 0.5
        Kd = X @ X.mT
        d = torch.rand(B, N, generator=g, device=dev) + 0.5
        rhs = torch.randn(B, N, c, generator=g, device=dev)
        desc = K.dense_diag_descriptor(Kd, d)
        dense = Kd.double() + torch.diag_embed(d.double())
        darg, const = d, False
        n_d += 1
    pre = None
    if k:
        L, _ = K.pivoted_cholesky(desc, k, contiguous=False)
        pre = K.precond_build(L, darg, const)
    ntri = rnd.choice([0, min(c, 16)])
    res = K.cg_solve(desc, rhs, precond=pre, n_tridiag=ntri, tolerance=1e-5, max_iter=1500)
    exact = torch.linalg.solve(dense, rhs.double())
    err = ((res.x.double() - exact).norm(dim=-2) / exact.norm(dim=-2)).max().item()
    assert err < 2e-3 and not res.nan_detected, (desc.kind, B, N, c, k, ntri, res.iterations, err)
    y = K.matvec(desc, rhs)
    merr = ((y.double() - dense @ rhs.double()).norm(dim=-2) / (dense @ rhs.double()).norm(dim=-2)).max().item()
    assert merr < 1e-5, ("matvec", desc.kind, B, N, c, merr)
print(f"fuzz ok: {n_k} Kronecker, {n_d} dense operators, seed {args.seed}")
