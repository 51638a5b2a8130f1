#!/bin/bash
python tools/mb_pc_sweep.py 2>&1 | grep "^R="
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^E " gpurun_out/pytest_gpu.log | head -5
LO_OC_GW8=1 python tools/mb_pc_sweep.py 2>&1 | grep "^R=" | head -4
