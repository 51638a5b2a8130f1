#!/bin/bash
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^E " gpurun_out/pytest_gpu.log | head -5
for i in 1 2; do python tools/mb_cfg45.py cfg4 2>&1 | grep -E "CG:|precond_fused"; done
