cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "root_form or precond" 2>&1 | tail -5
timeout 300 python tools/mb_e2e.py 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from linear_operator_amd import _hip, kernels as K
B, N, R = 512, 8192, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
Cm = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
d = torch.rand(B, N, generator=g, device="cuda") + 0.5
desc = K.lowrank_diag_descriptor(Cm, d)
L, perm = K.pivoted_cholesky(desc, 15, contiguous=False)
_hip.prof_enable(True)
for _ in range(3): K.precond_build(L, d, False, root=Cm, perm=perm, need_q=False)
torch.cuda.synchronize()
for k, (c, ms) in sorted(_hip.prof_report().items()): print(f"    {k:20s} {c:5d} x {ms / c * 1e3:9.1f} us")
PY
