"""Stress of the one-launch solve: many repetitions of A.solve(rhs) (cfg2 and the headline batch) interleaved with
allocator churn; every result must be bit-identical to the first one (the kernel is deterministic by construction)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import kernels as K, settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
N, R = 8192, 32
reps = int(os.environ.get("STRESS_REPS", 1500))
g = torch.Generator(device="cuda"); g.manual_seed(77)
bad = 0
for B in (64, 512, 70):
    Cm = torch.randn(B, N, R, generator=g, device="cuda") / (R ** 0.5)
    d = torch.rand(B, N, generator=g, device="cuda") + 0.5
    rhs = torch.randn(B, N, 1, generator=g, device="cuda")
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
    first = None
    t0 = time.perf_counter()
    with settings.cg_tolerance(1e-4):
        for i in range(reps if B != 512 else reps // 4):
            clear_preconditioner_memo()
            if i % 7 == 3:  # allocator churn: blocks of other sizes come and go between the solves
                junk = [torch.empty(int(s), device="cuda").normal_() for s in (1e5, 3e6, 7e4)]
                del junk
            K._hip.prof_enable(True)
            x = A.solve(rhs)
            names = set(K._hip.prof_report()); K._hip.prof_enable(False)
            if first is None:
                first = x.clone()
            if names != {"solve_fused"} or not torch.equal(x, first):
                bad += 1
                print(f"B={B} rep {i}: kernels {sorted(names)}, finite {torch.isfinite(x).all().item()}, "
                      f"max diff {(x - first).abs().max().item():.3e}", flush=True)
                if bad > 10: sys.exit(1)
    torch.cuda.synchronize()
    print(f"B={B}: {reps if B != 512 else reps // 4} solves in {time.perf_counter() - t0:.1f} s, failures so far {bad}", flush=True)
print("stress done, failures:", bad)
