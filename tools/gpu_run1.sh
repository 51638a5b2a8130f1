cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x > gpurun_out/t_dist.log 2>&1
tail -40 gpurun_out/t_dist.log
