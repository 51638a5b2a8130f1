cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -k "lanczos or root_decomp or diagonalization or sqrt_inv or minres" 2>&1 | tail -15
timeout 300 python tools/mb_lanczos.py 2>&1 | grep -v amdgpu.ids
LO_LZ_UNFUSED=1 timeout 300 python tools/mb_lanczos.py 2>&1 | grep -v amdgpu.ids | head -4
