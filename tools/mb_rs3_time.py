"""Launch time of the headline kernel k_cg_rspace3 (diagonal-form R-space solve, 512 x 8192 x 32, one column); a variant
library when LO_LIB_VARIANT names one (tools/build_variant.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LO_EIGFORM_AFTER_USES", "0")
from linear_operator_amd import _hip, kernels as K
if os.environ.get("LO_LIB_VARIANT"):
    _hip._LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants",
                                  "liblo_amd_%s.so" % os.environ["LO_LIB_VARIANT"])
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
d = torch.rand(B, N, generator=g, device=dev) + 0.5
rhs = torch.randn(B, N, 1, generator=g, device=dev)
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(K.lowrank_diag_descriptor(Cm, None), 15, contiguous=False)
pre = K.precond_build(L, d, False, root=Cm, perm=perm)
for _ in range(5): K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
torch.cuda.synchronize()
_hip.prof_enable(True)
for _ in range(100): K.cg_solve(desc, rhs, precond=pre, tolerance=1e-4)
torch.cuda.synchronize()
rep = _hip.prof_report(); _hip.prof_enable(False)
print(os.environ.get("LO_LIB_VARIANT", "tree"), " ".join("%s %.1f us" % (k, 1e3 * v[1] / v[0]) for k, v in sorted(rep.items())), K.cg_last_executed())
