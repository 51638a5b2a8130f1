#!/bin/bash
python -m pytest tests/test_gpu_sweep.py -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -30 gpurun_out/pytest_gpu.log
