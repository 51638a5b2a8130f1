"""Where the per-step cost of the overlapped all-gather goes (1-rank RCCL group on one GPU): solve alone, solve + gather
issued inline, solve + gather issued by a helper thread, with / without LO_OC_RESERVE_CUS."""
import os, sys, time, threading, queue, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
from linear_operator_amd import kernels as K
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
Cm, d, rhs = bench.make_problem(dev, 1234)
desc = K.lowrank_diag_descriptor(Cm, d); pre = bench.build_precond(desc, d)
bufs = [torch.empty(512, 8192, 1, device=dev) for _ in range(2)]
big_src = torch.empty(7 * 512, 8192, 1, device=dev)   # what 7 peers would write into this rank per step at 8 ranks
big_dst = [torch.empty(7 * 512, 8192, 1, device=dev) for _ in range(2)]
def run(mode, steps=40):
    pend = []
    def solve(): return K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
    for _ in range(3): solve()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        r = solve()
        if mode == "inline":
            while len(pend) >= 2: pend.pop(0)[0].wait()
            pend.append((dist.all_gather_into_tensor(bufs[k % 2], r.x, async_op=True), r.x))
        elif mode == "copy":  # same bytes moved by a plain device copy on a side stream
            with torch.cuda.stream(side):
                side.wait_stream(torch.cuda.current_stream()); bufs[k % 2].copy_(r.x)
        elif mode == "copy8":  # the HBM side of an 8-rank gather: 112 MB written per step (here also read) on a side stream
            with torch.cuda.stream(side):
                side.wait_stream(torch.cuda.current_stream()); big_dst[k % 2].copy_(big_src)
    for w, _ in pend: w.wait()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
side = torch.cuda.Stream()
for reserve in ("0", "32"):
    os.environ["LO_OC_RESERVE_CUS"] = reserve
    print(f"reserve {reserve}: solve only {run('none'):.3f} ms | + inline gather {run('inline'):.3f} ms | + side-stream copy {run('copy'):.3f} ms | + 112 MB side-stream copy per step (8-rank inbound volume) {run('copy8'):.3f} ms")
dist.destroy_process_group()
