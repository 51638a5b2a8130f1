cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_api.py -q -m gpu > gpurun_out/t_api.log 2>&1
tail -30 gpurun_out/t_api.log
