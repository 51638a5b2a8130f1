"""Where the host time of A.solve(rhs) on the fused path goes (64 members: the kernel is 0.15 ms): the bare C call with
everything preallocated, kernels.solve_fused, the pieces of the operator path, the whole API call."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K, settings
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo
import importlib; lcg_mod = importlib.import_module("linear_operator_amd.utils.linear_cg")
B, N, R = int(os.environ.get("FU_B", 64)), 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))
desc = K.lowrank_diag_descriptor(Cm, d)
lib = _hip.load()
def timeit(fn, reps=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
# bare C call
prm = K._cg_params(1, 0, 1000, 20, 1e-4, 1e-10, 1e-10, 0)
s = desc.c_struct()
x = torch.empty_like(rhs); small = torch.empty(B * (3 * R * R + 1 + 15), device="cuda")
dinv = torch.empty(B, N, device="cuda")
ws = _hip.workspace(lib.lo_solve_fused_workspace_bytes(C.byref(s), 15, C.byref(prm)), rhs.device)
info = _hip.FusedInfo(); st = _hip.stream_ptr(rhs.device)
p = [small.data_ptr() + 4 * o for o in (0, B * R * R, 2 * B * R * R, 3 * B * R * R, 3 * B * R * R + B)]
def bare():
    lib.lo_solve_fused_f32(C.byref(s), 15, 1e-3, C.byref(prm), rhs.data_ptr(), x.data_ptr(), p[0], p[1], p[2], dinv.data_ptr(),
                           p[3], p[4], ws.data_ptr(), ws.numel(), C.byref(info), st)
print(f"bare C call                 {timeit(bare):7.1f} us")
print(f"kernels.solve_fused         {timeit(lambda: K.solve_fused(desc, rhs, 15, 1e-3, tolerance=1e-4)):7.1f} us")
with settings.cg_tolerance(1e-4):
    def api():
        clear_preconditioner_memo()
        return A.solve(rhs)
    K._hip.prof_enable(True)
    print(f"A.solve(rhs)                {timeit(api):7.1f} us")
    torch.cuda.synchronize(); pr = K._hip.prof_report(); K._hip.prof_enable(False)
    print("   kernel avg inside that loop:", {k: round(v[1] / v[0] * 1e3, 1) for k, v in pr.items()})
    def pre_only():
        clear_preconditioner_memo()
        return A._solve_preconditioner()
    print(f"_solve_preconditioner alone {timeit(pre_only):7.1f} us")
    print(f"_kernel_descriptor          {timeit(A._kernel_descriptor):7.1f} us")
    print(f"clear memo                  {timeit(clear_preconditioner_memo):7.1f} us")
    pc = pre_only()
    print(f"lower closure               {timeit(lambda: lcg_mod._lower_matmul_closure(A._matmul, rhs.shape[:-2])):7.1f} us")
with settings.cg_tolerance(1e-4):
    K._hip.prof_enable(True); api(); torch.cuda.synchronize(); print("kernels of one A.solve:", sorted(K._hip.prof_report())); K._hip.prof_enable(False)
