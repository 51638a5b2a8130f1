#!/bin/bash
python tools/mb_resident_sweep.py 2>&1 | grep "^R=" | grep -E "N=  (1024|2048|4096|8192)"
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^E " gpurun_out/pytest_gpu.log | head -5
