"""Headline-shape batched matvec (512 x 8192 x 32, c columns) in a loop: for rocprofv3 --kernel-trace --stats and the PMC passes.
   python tools/mb_lowrank_mv.py [c] [reps] [two-pass]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import kernels  # noqa: E402

c = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if len(sys.argv) > 3 and sys.argv[3] == "two-pass":
    os.environ["LO_NO_RESIDENT_MV"] = "1"
B, N, R = 512, 8192, 32
g = torch.Generator().manual_seed(1234)
C = (torch.randn(B, N, R, generator=g) / R ** 0.5).cuda()
d = (torch.rand(B, N, generator=g) + 0.5).cuda()
v = torch.randn(B, N, c, generator=g).cuda()
desc = kernels.lowrank_diag_descriptor(C, d)
for _ in range(30):
    kernels.matvec(desc, v)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = kernels.matvec(desc, v)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
by = 4 * B * (N * R + N + 2 * N * c)
print(f"c={c}: {us:.1f} us per batched matvec (wall) = {by / us / 1e6:.2f} TB/s algorithmic = {by / us / 8e6:.3f} of 8 TB/s")
