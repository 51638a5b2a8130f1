"""Host cost of the headline step: resident kernel (HIP events) / bare C call with everything prebuilt / kernels.cg_solve."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
rhs = torch.randn(B, N, 1, generator=g, device="cuda")
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(desc, 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)
lib = _hip.load()
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
s = desc.c_struct(); ps = pre.c_struct()
prm = K._cg_params(1, 0, 1000, 20, 1e-4, 1e-10, 1e-10, 0)
ws = _hip.workspace(lib.lo_cg_workspace_bytes(C.byref(s), C.byref(ps), C.byref(prm)), rhs.device)
x = torch.empty_like(rhs); info = _hip.CgInfo(); st = _hip.stream_ptr(rhs.device); cb = _hip.MATVEC_CB()
def bare():
    lib.lo_cg_solve_f32(C.byref(s), cb, None, C.byref(ps), cb, None, C.byref(prm), rhs.data_ptr(), None, x.data_ptr(), None,
                        ws.data_ptr(), ws.numel(), C.byref(info), st)
step = lambda: K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
K._hip.prof_enable(True)
for _ in range(10): step()
torch.cuda.synchronize(); pr = K._hip.prof_report(); K._hip.prof_enable(False)
print("kernel (HIP events)  ", {k: round(v[1] / v[0] * 1e3, 1) for k, v in pr.items()})
print(f"bare C call           {timeit(bare):7.1f} us")
print(f"kernels.cg_solve      {timeit(step):7.1f} us")
print(f"lo_cg_workspace_bytes {timeit(lambda: lib.lo_cg_workspace_bytes(C.byref(s), C.byref(ps), C.byref(prm))):7.1f} us")
print(f"c_struct x2 + params  {timeit(lambda: (desc.c_struct(), pre.c_struct(), K._cg_params(1, 0, 1000, 20, 1e-4, 1e-10, 1e-10, 0))):7.1f} us")
print(f"torch.empty x2        {timeit(lambda: (_hip.workspace(ws.numel(), rhs.device), torch.empty_like(rhs))):7.1f} us")
