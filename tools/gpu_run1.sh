#!/bin/bash
# scratch runner for one gpurun call
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python tools/mb_cfg45.py cfg4 2>&1 | grep -E "CG:|precond_fused|pivoted|update_p|kron"
