cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lockstep or onchip" > gpurun_out/t_lockstep.log 2>&1
tail -25 gpurun_out/t_lockstep.log
