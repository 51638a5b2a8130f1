"""Timeline of ONE cfg3 inv_quad_logdet (what bench.py's cfg3 extra times) from a rocprofv3 kernel trace: every kernel
with its start offset, duration and the idle gap in front of it.  Two modes:
  python tools/iql_timeline.py run | train    (the workload at the C ABI / forward + backward through the operator API;
                                               run it under `rocprofv3 --kernel-trace --output-format csv`)
  python tools/iql_timeline.py parse <kernel_trace.csv>
"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def run(mode):
    import torch
    from linear_operator_amd import kernels as K
    import bench
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(77)
    B, N, R = 512, 8192, 32
    Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    full = torch.randn(B, N, 17, generator=g, device=dev); full[..., :16] /= full[..., :16].norm(dim=-2, keepdim=True)
    desc = K.lowrank_diag_descriptor(Cm, d)

    def iql():
        pre = bench.build_precond(desc, d)
        r = K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=1e-4)
        _, _, ld = K.tridiag_eigh_slq(r.t_mat, N)
        return r, ld + pre.logdet
    if mode == "run":
        for _ in range(6):
            iql()
        torch.cuda.synchronize()
        return
    # the host API, forward + backward (bench.py's cfg3_..._forward_backward_host_api)
    from linear_operator_amd import settings as lo_settings
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
    Cg, dg = Cm.clone().requires_grad_(True), d.clone().requires_grad_(True)
    y = full[..., 16:].contiguous()
    with lo_settings.cg_tolerance(1e-4), lo_settings.num_trace_samples(16):
        for _ in range(6):
            clear_preconditioner_memo()
            Cg.grad = dg.grad = None
            A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
            iq, ld = A.inv_quad_logdet(y, logdet=True)
            (iq.sum() + ld.sum()).backward()
    torch.cuda.synchronize()


def parse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last occurrence of the factorisation kernel starts the last iteration
    first = max(i for i, r in enumerate(rows) if "k_pc_onchip4" in r["Kernel_Name"])
    while first > 0 and "zero_span" in rows[first - 1]["Kernel_Name"]:
        first -= 1
    t0 = int(rows[first]["Start_Timestamp"]); prev_end = t0; busy = 0
    for r in rows[first:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][-60:]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
        busy += e - s; prev_end = e
    print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(prev_end - t0 - busy) / 1e3:.1f} us")


if __name__ == "__main__":
    run(sys.argv[1]) if sys.argv[1] in ("run", "train") else parse(sys.argv[2])
