cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python tools/mb_lockstep.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/mb_lockstep.log
LS_C=1 timeout 300 python tools/mb_lockstep.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/mb_lockstep.log
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 >> gpurun_out/mb_lockstep.log
cat gpurun_out/mb_lockstep.log
