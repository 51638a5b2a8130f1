"""The gather of SURVEY 8(e) as peer writes, emulated on ONE GPU: the x pass of the resident headline
solve (k_cg_rspace3) stores every solution value into n more buffers -- on a node: the IPC-mapped gather buffers of the
peers, written over each peer's own xGMI link while the solve runs; here: n local buffers (the store instructions and their
HBM traffic, not the links).  Prints ms per solve for n = 0 .. 7 and checks that every buffer holds the solution."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import kernels as K  # noqa: E402

os.environ.setdefault("LO_EIGFORM_AFTER_USES", "0")
B, N, R = 512, 8192, 32
dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(3)
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
rhs = torch.randn(B, N, 1, generator=g, device=dev)
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm, need_q=False)
world, rank = 8, 3  # this rank's members sit at [rank * B, (rank + 1) * B) of every peer's buffer
peers = [torch.zeros(world * B, N, device=dev) for _ in range(7)]


def timed(n, reps=200):
    K.peer_gather_set(peers[:n], rank * B)
    for _ in range(30):
        r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


base, r0 = timed(0)
print(f"engine: {K.cg_last_executed()['rspace']}, diagonal form {K.cg_last_executed()['rspace_diag']}")
print(f"n = 0 peers: {base:.4f} ms per solve")
for n in (1, 3, 7):
    for p in peers:
        p.zero_()
    ms, r = timed(n)
    ok = all(torch.equal(peers[i][rank * B:(rank + 1) * B], r.x[..., 0]) for i in range(n))
    untouched = all(float(peers[i][:rank * B].abs().max()) == 0.0 for i in range(n))
    print(f"n = {n} peers: {ms:.4f} ms per solve (+{(ms - base) * 1e3:.1f} us, {n * B * N * 4 / 1e6:.0f} MB of extra stores), "
          f"buffers hold the solution: {ok}, other slices untouched: {untouched}")
K.peer_gather_set(())
