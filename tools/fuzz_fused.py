"""Randomised shapes through the one-launch solve (lo_solve_fused_f32): pivots against the resident pivoted Cholesky
(itself bit-exact against the oracle: tools/fuzz_resident.py), the solution against the exact fp64 Woodbury solution,
the root form against the three-launch build.  FUZZ_SECONDS bounds the run (default 120)."""
import os, random, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K
rng = random.Random(int(os.environ.get("FUZZ_SEED", 3)))
t_end = time.time() + float(os.environ.get("FUZZ_SECONDS", 120))
n = bad = fallbacks = 0
while time.time() < t_end:
    B = rng.choice([1, 2, 7, 33, 64, 65, 130, 300])
    N = rng.choice([256, 300, 1000, 1024, 1025, 2048, 3000, 4096, 4097, 6000, 8192, 9000, 12000, 16384])
    R = rng.choice([8, 16, 32]); c = rng.randint(1, 4)
    if B * N * R > 5e7: continue
    const = rng.random() < 0.3
    g = torch.Generator(device="cuda"); g.manual_seed(rng.randrange(1 << 30))
    Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    d = (torch.rand(B, generator=g, device="cuda") + 0.5) if const else (torch.rand(B, N, generator=g, device="cuda") + 0.5)
    rhs = torch.randn(B, N, c, generator=g, device="cuda")
    rank = min(15, R) if R >= 15 else R
    desc = K.lowrank_diag_descriptor(Cm, d, const_diag=const)
    if not K.solve_fused_supported(desc, c, rank): continue
    out = K.solve_fused(desc, rhs, rank, 1e-3, tolerance=1e-4)
    n += 1
    if out is None:
        fallbacks += 1; continue
    Lr, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), rank, 1e-3)
    pe = torch.equal(out.permutation(N)[:, :rank], perm[:, :rank]) if Lr.shape[-1] == rank else True
    d2 = (d[:, None].expand(B, N) if const else d).double().unsqueeze(-1)
    C64 = Cm.double(); Cd = C64 / d2
    cap = torch.eye(R, device="cuda", dtype=torch.float64) + C64.mT @ Cd
    xs = rhs.double() / d2 - Cd @ torch.linalg.solve(cap, C64.mT @ (rhs.double() / d2))
    err = ((out.cg.x.double() - xs).norm(dim=-2) / xs.norm(dim=-2)).max().item()
    if not pe or not err < 1e-4:
        bad += 1; print(f"FAIL B={B} N={N} R={R} c={c} const={const}: pivots equal {pe}, rel err {err:.2e}", flush=True)
print(f"fuzz_fused: {n} solves ({fallbacks} fell back to the three-launch path by status), {bad} failures")
sys.exit(1 if bad else 0)
