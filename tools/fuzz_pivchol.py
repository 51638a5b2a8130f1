"""Random shapes through the resident pivoted Cholesky (k_pc_onchip4) against the streaming engine (lo_pivchol.hip) of the
same library: permutation and factor bit for bit (both follow the oracle's operation order).  Shapes: every group size
(1 .. 32 workgroups per member), ragged N, root ranks 1 .. 32 (padded to 8 / 16 / 32), ranks 1 .. 32 (64 KB and 128 KB
of L rows per workgroup), rank-deficient members.  usage: python tools/fuzz_pivchol.py [--minutes M] [--seed S]"""
import argparse, os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import _hip, kernels as K

ap = argparse.ArgumentParser(); ap.add_argument("--minutes", type=float, default=2.0); ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rnd = random.Random(args.seed)
g = torch.Generator(device="cuda"); g.manual_seed(args.seed)
t_end = time.time() + 60 * args.minutes
n = 0; took = 0; worst = None
while time.time() < t_end:
    N = rnd.choice([256, 257, 300, 1000, 1024, 1025, 2048, 3000, 4097, 8192, 8193, 12000, 16384, 20000, 32768, rnd.randint(256, 32768)])
    R = rnd.choice([1, 3, 8, 9, 15, 16, 17, 27, 32, rnd.randint(1, 32)])
    rank = min(N, rnd.choice([1, 2, 5, 8, 9, 15, 16, 17, 24, 31, 32, rnd.randint(1, 32)]))
    B = rnd.choice([1, 2, 7, 33, 64, 130, rnd.randint(1, 300)])
    B = max(1, min(B, (1 << 27) // (N * 32)))
    C = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    if rnd.random() < 0.2:   # rank-deficient members / repeated rows (ties)
        C[:, N // 2:] = C[:, : N - N // 2].clone()
    if rnd.random() < 0.1:
        C[0] = 0.0
    desc = K.lowrank_diag_descriptor(C, None)
    _hip.prof_enable(True)
    L, piv = K.pivoted_cholesky(desc, rank)
    torch.cuda.synchronize()
    prof = _hip.prof_report(); _hip.prof_enable(False)
    took += "pc_onchip" in prof
    _hip.load().lo_cg_set_onchip(0)
    try:
        L0, piv0 = K.pivoted_cholesky(desc, rank)
    finally:
        _hip.load().lo_cg_set_onchip(1)
    same = torch.equal(piv, piv0) and torch.equal(torch.nan_to_num(L, nan=12345.0), torch.nan_to_num(L0, nan=12345.0))
    if not same:
        print(f"MISMATCH B={B} N={N} R={R} rank={rank} seed={args.seed} case={n}: pivots equal {torch.equal(piv, piv0)}", flush=True)
        sys.exit(1)
    n += 1
print(f"fuzz_pivchol ok: {n} cases ({took} on the resident kernel), permutation and factor bit-identical to the streaming engine")
