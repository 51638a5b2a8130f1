#!/bin/bash
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -1; grep -E "^E " gpurun_out/pytest_gpu.log | head -6
python tools/mb_cfg45.py cfg4 2>&1 | grep -E "CG:|kron_gemm"
