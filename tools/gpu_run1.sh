#!/bin/bash
python -m pytest tests/test_gpu_sweep.py -q -x -k small_members > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^E " gpurun_out/pytest_gpu.log | head -8
