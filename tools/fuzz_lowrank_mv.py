"""Fuzz of the one-pass resident low-rank + diagonal matvec (csrc/lo_lowrank_mv.hip) through lo_matvec_f32: random batch
sizes, member sizes 256 ... 32768 (every group size, ragged tails), ranks 1 ... 32, 1 ... 4 columns, the three diagonal modes,
scaled / zero / duplicated columns of C; every case against float64, against itself (bitwise) and against the two-pass
kernels; every 16th case with all workgroups forced down the lost-hand-off path.  usage: fuzz_lowrank_mv.py [seconds] [seed]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import kernels as K  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g).item())


t_end = time.time() + budget
n = bad = resident = 0
worst = 0.0
while time.time() < t_end:
    N = [ri(256, 1100), ri(1000, 4200), ri(4000, 9000), ri(8000, 33000), 1024 * ri(1, 32), 256 * ri(1, 128)][ri(0, 5)]
    N = min(N, 32768)
    R = [ri(1, 32), 8, 16, 32, ri(17, 32)][ri(0, 4)]
    c = ri(1, 4)
    B = max(1, min(ri(1, 700), (96 << 20) // (N * max(R, 8) * 4)))
    C = torch.randn(B, N, R, generator=g) / R ** 0.5
    kind = ri(0, 5)
    if kind == 1:
        C[:, :, ri(0, R - 1)] *= 1e-4
    elif kind == 2 and R > 1:
        C[:, :, ri(0, R - 1)] = C[:, :, 0]
    elif kind == 3:
        C[ri(0, B - 1)] = 0.0
    dmode = ri(0, 2)
    d = None if dmode == 0 else ((torch.rand(B, N, generator=g) + 10.0 ** -ri(0, 3)) if dmode == 1 else (torch.rand(B, generator=g) + 0.1))
    v = torch.randn(B, N, c, generator=g)
    Cd, vd = C.cuda(), v.cuda()
    dd = None if d is None else d.cuda()
    desc = K.lowrank_diag_descriptor(Cd, dd, const_diag=dmode == 2)
    C64, v64 = Cd.double(), vd.double()
    want = C64 @ (C64.mT @ v64)
    if dd is not None:
        want = want + (dd.double().reshape(-1, 1, 1) if dmode == 2 else dd.double().unsqueeze(-1)) * v64
    scale = (C64.abs() @ (C64.abs().mT @ v64.abs())).norm(dim=-2) + (0 if dd is None else 1) * want.norm(dim=-2) + 1e-300
    fb = n % 16 == 15
    if fb:
        os.environ["LO_MV_TEST_FALLBACK"] = "1"
    y = K.matvec(desc, vd)
    y2 = K.matvec(desc, vd)
    os.environ.pop("LO_MV_TEST_FALLBACK", None)
    os.environ["LO_NO_RESIDENT_MV"] = "1"
    y3 = K.matvec(desc, vd)
    del os.environ["LO_NO_RESIDENT_MV"]
    err = float(((y.double() - want).norm(dim=-2) / scale).max())
    err3 = float(((y3.double() - want).norm(dim=-2) / scale).max())
    rep = bool(torch.equal(y, y2))
    worst = max(worst, err)
    n += 1
    if not (err < 2e-6 and rep and err3 < 2e-6):
        bad += 1
        print(f"BAD B={B} N={N} R={R} c={c} dmode={dmode} kind={kind} fallback={fb}: err {err:.3e} two-pass {err3:.3e} reproducible {rep}", flush=True)
print(f"{n} cases, {bad} bad, worst error (relative to the size of the terms) {worst:.2e}, seed {seed}")
sys.exit(1 if bad else 0)
