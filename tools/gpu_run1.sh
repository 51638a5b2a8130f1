cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k cfg5 2>&1 | tail -15
