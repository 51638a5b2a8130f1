"""One-launch solve at N = 16384 (groups of 16 workgroups, round 4) against the three-launch path, operator API."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(5)
for B, N in ((256, 16384), (512, 12000), (512, 8192)):
    Cm = torch.randn(B, N, 32, generator=g, device=dev) / 32 ** 0.5
    d = torch.rand(B, N, generator=g, device=dev) + 0.5
    rhs = torch.randn(B, N, 1, generator=g, device=dev)
    def run():  # (a new operator every time, as in a training loop: nothing cached on the object or in the memo)
        clear_preconditioner_memo()
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
        return A.solve(rhs)
    for tag, env in (("one launch", {}), ("three launches", {"LO_NO_FUSED_SOLVE": "1"})):
        os.environ.update(env)
        with settings.cg_tolerance(1e-4):
            for _ in range(2): x = run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): x = run()
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
        for k in env: del os.environ[k]
        print(f"B={B} N={N}: {tag:14s} {t * 1e3:.3f} ms")
