"""One-pass resident low-rank + diagonal matvec (csrc/lo_lowrank_mv.hip) against float64 and against the two-pass kernels;
timing of both at the headline shape.  Run on the GPU box:  python tools/check_lowrank_mv.py [--time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import kernels  # noqa: E402


def ref64(C, d, v, const):
    C64, v64 = C.double(), v.double()
    y = C64 @ (C64.transpose(-1, -2) @ v64)
    if d is not None:
        y = y + (d.double().reshape(-1, 1, 1) if const else d.double().unsqueeze(-1)) * v64
    return y


def one(B, N, R, c, diag, seed=0):
    g = torch.Generator().manual_seed(seed)
    C = (torch.randn(B, N, R, generator=g) / R ** 0.5).cuda()
    v = torch.randn(B, N, c, generator=g).cuda()
    d = None
    if diag == "full":
        d = (torch.rand(B, N, generator=g) + 0.5).cuda()
    elif diag == "const":
        d = (torch.rand(B, generator=g) + 0.5).cuda()
    desc = kernels.lowrank_diag_descriptor(C, d, const_diag=diag == "const")
    want = ref64(C, d, v, diag == "const")
    out = {}
    for mode in ("resident", "fallback", "two-pass"):
        os.environ.pop("LO_NO_RESIDENT_MV", None)
        os.environ.pop("LO_MV_TEST_FALLBACK", None)
        if mode == "two-pass":
            os.environ["LO_NO_RESIDENT_MV"] = "1"
        if mode == "fallback":
            os.environ["LO_MV_TEST_FALLBACK"] = "1"
        y = kernels.matvec(desc, v)
        y2 = kernels.matvec(desc, v)
        torch.cuda.synchronize()
        err = ((y.double() - want).norm(dim=-2) / want.norm(dim=-2)).max().item()
        out[mode] = (err, torch.equal(y, y2))
    os.environ.pop("LO_NO_RESIDENT_MV", None)
    os.environ.pop("LO_MV_TEST_FALLBACK", None)
    ok = all(e < 2e-6 and rep for e, rep in out.values())
    print(f"B={B:5d} N={N:6d} R={R:3d} c={c} diag={diag:5s} " +
          "  ".join(f"{m}: {e:.2e}{'' if rep else ' NOT-REPRODUCIBLE'}" for m, (e, rep) in out.items()) +
          ("" if ok else "   <-- FAIL"), flush=True)
    return ok


def timing():
    B, N, R = 512, 8192, 32
    g = torch.Generator().manual_seed(1234)
    C = (torch.randn(B, N, R, generator=g) / R ** 0.5).cuda()
    d = (torch.rand(B, N, generator=g) + 0.5).cuda()
    for c in (1, 2):
        v = torch.randn(B, N, c, generator=g).cuda()
        desc = kernels.lowrank_diag_descriptor(C, d)
        for mode in ("resident", "two-pass"):
            if mode == "two-pass":
                os.environ["LO_NO_RESIDENT_MV"] = "1"
            else:
                os.environ.pop("LO_NO_RESIDENT_MV", None)
            for _ in range(20):
                kernels.matvec(desc, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 200
            e0.record()
            for _ in range(reps):
                kernels.matvec(desc, v)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            bytes_ = 4 * B * (N * R + N + 2 * N * c)
            print(f"c={c} {mode:9s}: {us:7.1f} us per batched matvec (wall, incl. launch) -> {bytes_ / us / 1e6:.2f} TB/s "
                  f"= {bytes_ / us / 1e6 / 8:.3f} of 8 TB/s", flush=True)
        os.environ.pop("LO_NO_RESIDENT_MV", None)
    for wpc in (1, 2, 3):
        os.environ["LO_MV_WGS_PER_CU"] = str(wpc)
        v = torch.randn(B, N, 1, generator=g).cuda()
        desc = kernels.lowrank_diag_descriptor(C, d)
        for _ in range(20):
            kernels.matvec(desc, v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            kernels.matvec(desc, v)
        e1.record()
        torch.cuda.synchronize()
        print(f"c=1 resident, {wpc} workgroup(s) per CU: {e0.elapsed_time(e1) * 5:.1f} us", flush=True)
    os.environ.pop("LO_MV_WGS_PER_CU", None)
    os.environ["LO_MV_DEBUG"] = "1"
    kernels.matvec(desc, v)
    torch.cuda.synchronize()
    os.environ.pop("LO_MV_DEBUG", None)


if __name__ == "__main__":
    ok = True
    for (B, N, R, c, diag) in [
        (64, 8192, 32, 1, "full"), (64, 8192, 32, 2, "full"), (7, 8192, 32, 1, "const"), (5, 8192, 32, 1, "none"),
        (33, 5000, 32, 1, "full"), (33, 5001, 32, 2, "full"), (9, 300, 32, 1, "full"), (9, 1024, 32, 1, "full"),
        (9, 1025, 32, 1, "full"), (3, 20000, 32, 1, "full"), (3, 32768, 32, 2, "const"), (40, 4096, 16, 1, "full"),
        (40, 4097, 16, 4, "full"), (40, 3000, 16, 3, "const"), (40, 2048, 8, 1, "full"), (40, 2500, 8, 4, "full"),
        (40, 8192, 8, 2, "none"), (600, 2048, 32, 1, "full"), (1000, 700, 24, 1, "full"), (300, 1500, 5, 2, "full"),
    ]:
        ok = one(B, N, R, c, diag) and ok
    print("ALL OK" if ok else "FAILURES", flush=True)
    if "--time" in sys.argv:
        timing()
    sys.exit(0 if ok else 1)
