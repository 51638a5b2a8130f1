"""Launch time of the resident pivoted Cholesky (k_pc_onchip4) at the headline shape: 512 x 8192 x 32, rank 15; optional
argument: N (members keep 32 columns)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import _hip, kernels as K
if os.environ.get("LO_LIB_VARIANT"):  # experiments: variants/liblo_amd_<name>.so (tools/build_variant.sh)
    _hip._LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants",
                                  "liblo_amd_%s.so" % os.environ["LO_LIB_VARIANT"])
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
RANK = int(sys.argv[2]) if len(sys.argv) > 2 else 15
B, R = 512 * 8192 // N, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
desc = K.lowrank_diag_descriptor(Cm, None)
for _ in range(5): K.pivoted_cholesky(desc, RANK, contiguous=False)
torch.cuda.synchronize()
_hip.prof_enable(True)
for _ in range(20): K.pivoted_cholesky(desc, RANK, contiguous=False)
torch.cuda.synchronize()
rep = _hip.prof_report()
_hip.prof_enable(False)
print(os.environ.get("LO_LIB_VARIANT", "tree"), "N", N, "rank", RANK, " ".join("%s %.1f us" % (k, 1e3 * v[1] / v[0]) for k, v in sorted(rep.items())))
