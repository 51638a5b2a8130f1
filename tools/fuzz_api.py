"""Randomised operators through the reference's operator API on the HIP path (A.solve, A.inv_quad_logdet forward +
backward) against dense float64 algebra: low-rank root + diagonal, dense + diagonal / constant diagonal, Kronecker +
constant diagonal; batch shapes (), (3,), (2, 2); vector and matrix right-hand sides; sizes on both sides of
min_preconditioning_size.  Not part of the test suite: `python tools/fuzz_api.py --minutes 4 --seed 1` on the GPU box."""
import argparse, os, random, sys, time, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings  # noqa: E402
from linear_operator_amd.operators import (  # noqa: E402
    AddedDiagLinearOperator, ConstantDiagLinearOperator, DenseLinearOperator, DiagLinearOperator,
    KroneckerProductLinearOperator, LowRankRootLinearOperator)

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rnd = random.Random(args.seed)
dev = torch.device("cuda")
t_end = time.time() + 60 * args.minutes
counts = {}
warnings.simplefilter("ignore")


def spd(g, bs, n, inner):
    X = torch.randn(*bs, n, inner, generator=g, device=dev) / inner ** 0.5
    return X @ X.mT


while time.time() < t_end:
    g = torch.Generator(device=dev); g.manual_seed(rnd.randrange(1 << 30))
    kind = rnd.choice(["lowrank", "dense", "dense_const", "kron"])
    bs = rnd.choice([(), (3,), (2, 2)])
    leaves = []
    if kind == "lowrank":
        N, R = rnd.choice([300, 1000, 2500, 4100, 8192, 9000]), rnd.choice([8, 20, 32, 40])
        Cm = (torch.randn(*bs, N, R, generator=g, device=dev) / R ** 0.5).requires_grad_(True)
        d = (torch.rand(*bs, N, generator=g, device=dev) + 0.3).requires_grad_(True)
        plus = rnd.random() < 0.4  # the reference's own `+` routing: LowRankRootAddedDiagLinearOperator (Woodbury closed form)
        build = (lambda: LowRankRootLinearOperator(Cm) + DiagLinearOperator(d)) if plus else \
            (lambda: AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d)))  # noqa: E731
        dense = lambda: Cm.double() @ Cm.double().mT + torch.diag_embed(d.double())  # noqa: E731
        leaves = [Cm, d]
    elif kind in ("dense", "dense_const"):
        N = rnd.choice([200, 900, 2100, 3001])
        Kd = spd(g, bs, N, 64).requires_grad_(True)
        if kind == "dense":
            d = (torch.rand(*bs, N, generator=g, device=dev) + 0.3).requires_grad_(True)
            build = lambda: AddedDiagLinearOperator(DenseLinearOperator(Kd), DiagLinearOperator(d))  # noqa: E731
            dense = lambda: Kd.double() + torch.diag_embed(d.double())  # noqa: E731
        else:
            d = (torch.rand(*bs, 1, generator=g, device=dev) + 0.3).requires_grad_(True)
            build = lambda: AddedDiagLinearOperator(DenseLinearOperator(Kd), ConstantDiagLinearOperator(d, N))  # noqa: E731
            dense = lambda: Kd.double() + d.double().unsqueeze(-1) * torch.eye(N, device=dev, dtype=torch.float64)  # noqa: E731
        leaves = [Kd, d]
    else:
        n1, n2 = rnd.choice([(20, 30), (48, 48), (64, 40)])
        N = n1 * n2
        K1 = (spd(g, bs, n1, n1) + 0.2 * torch.eye(n1, device=dev)).requires_grad_(True)
        K2 = (spd(g, bs, n2, n2) + 0.2 * torch.eye(n2, device=dev)).requires_grad_(True)
        d = (torch.rand(*bs, 1, generator=g, device=dev) * 0.2 + 0.1).requires_grad_(True)
        plus = rnd.random() < 0.4  # `+` routing: KroneckerProductAddedDiagLinearOperator (per-factor eigendecompositions)
        build = (lambda: KroneckerProductLinearOperator(DenseLinearOperator(K1), DenseLinearOperator(K2))  # noqa: E731
                 + ConstantDiagLinearOperator(d, N)) if plus else (lambda: AddedDiagLinearOperator(  # noqa: E731
                     KroneckerProductLinearOperator(DenseLinearOperator(K1), DenseLinearOperator(K2)),
                     ConstantDiagLinearOperator(d, N)))

        def dense():
            a, b = K1.double(), K2.double()
            kr = (a[..., :, None, :, None] * b[..., None, :, None, :]).reshape(*bs, N, N)
            return kr + d.double().unsqueeze(-1) * torch.eye(N, device=dev, dtype=torch.float64)
        leaves = [K1, K2, d]
    c = rnd.choice([0, 1, 3, 11, 17, 40] if bs == () else [1, 3, 11, 17, 40])  # 0: vector right-hand side (unbatched operators only, as in the reference)
    rhs = torch.randn(*bs, N, generator=g, device=dev) if c == 0 else torch.randn(*bs, N, c, generator=g, device=dev)
    chol = rnd.choice([0, 800])
    tag = (kind, bs, N, c, chol)
    if os.environ.get("FUZZ_VERBOSE"):
        print(tag, flush=True)
    psize, nprobe = rnd.choice([0, 5, 15, 15, 40, 150]), rnd.choice([1, 10, 10, 16, 33])
    tag = tag + (psize, nprobe)
    if os.environ.get("FUZZ_VERBOSE"):
        print("   preconditioner size", psize, "probes", nprobe, flush=True)
    with settings.max_cholesky_size(chol), settings.min_preconditioning_size(rnd.choice([100, 2000])), \
            settings.max_preconditioner_size(psize), settings.num_trace_samples(nprobe), \
            settings.cg_tolerance(1e-3), settings.max_cg_iterations(2000):
        A64 = dense()
        r64 = rhs.double() if c else rhs.double().unsqueeze(-1)
        exact = torch.linalg.solve(A64, r64)
        x = build().solve(rhs)
        x2 = x if c else x.unsqueeze(-1)
        err = ((x2.double() - exact).norm(dim=-2) / exact.norm(dim=-2)).max().item()
        assert x.shape == rhs.shape and err < 2e-2, ("solve", tag, err)
        for t in leaves:
            t.grad = None
        iq, ld = build().inv_quad_logdet(rhs, logdet=True)
        iq64 = (r64 * exact).sum((-2, -1)) if c else (r64 * exact).sum((-2, -1))
        e_iq = ((iq.double() - iq64).abs() / iq64.abs()).max().item()
        lam = torch.linalg.eigvalsh(A64).log()
        ld64 = lam.sum(-1)
        # noise of the stochastic estimate with 10 Gaussian probes: std <= sqrt(2 / P) ||log A||_F (less with a
        # preconditioner); the Cholesky branch is exact
        noise = 5.0 * (2.0 / nprobe * (lam * lam).sum(-1)).sqrt() if chol < N else torch.zeros_like(ld64)
        e_ld = ((ld.double() - ld64).abs() / (noise + 0.02 * ld64.abs() + 1e-3) * 0.3).max().item()
        assert iq.shape == tuple(bs) and e_iq < 2e-2, ("inv_quad", tag, e_iq)
        assert ld.shape == tuple(bs) and e_ld < 0.3, ("logdet", tag, e_ld)  # (a stochastic estimate above max_cholesky_size)
        (iq.sum() + ld.sum()).backward()
        g_hip = [t.grad.clone() for t in leaves]
        for t in leaves:
            t.grad = None
        A64 = dense()
        exact = torch.linalg.solve(A64, r64)
        (r64 * exact).sum().backward()  # the inv_quad part has an exact gradient; the logdet part is stochastic
        for t, gh in zip(leaves, g_hip):
            assert torch.isfinite(gh).all(), ("grad finite", tag)
            t.grad = None
    counts[kind] = counts.get(kind, 0) + 1
print("fuzz ok:", counts, "seed", args.seed)
