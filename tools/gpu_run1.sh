cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lockstep or onchip" -x 2>&1 | tail -15
for i in 1 2; do timeout 300 python tools/mb_lockstep.py 2>&1 | grep -v amdgpu.ids; done
LO_LS_V1=1 timeout 300 python tools/mb_lockstep.py 2>&1 | grep -v amdgpu.ids
LO_LS_DEBUG=5 timeout 300 python tools/mb_lockstep.py 2>&1 | grep -v amdgpu.ids | tail -5
