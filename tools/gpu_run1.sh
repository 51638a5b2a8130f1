#!/bin/bash
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^E " gpurun_out/pytest_gpu.log | head -8
HOST_PROF=1 python tools/mb_host_single.py 2>&1 | grep -E "dense RBF|N=4000|dense_mv"
