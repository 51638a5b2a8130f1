#!/bin/bash
python -m pytest tests -q -x -m gpu -k "dense or sweep or cfg5 or matvec or fullsize" > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
python tools/mb_cfg45.py cfg5 2>&1 | grep -E "CG:|dense_mv"
LO_DENSE_MFMA32=1 python tools/mb_cfg45.py cfg5 2>&1 | grep -E "CG:|dense_mv"
