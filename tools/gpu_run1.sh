#!/bin/bash
for n in 1024 2048 4096 8192; do LS_N=$n python tools/mb_lockstep.py 2>&1 | grep -E "cg_solve"; done
echo "== groups of 8 (old)"
for n in 1024 2048 4096; do LO_OC_GW8=1 LS_N=$n python tools/mb_lockstep.py 2>&1 | grep -E "cg_solve"; done
python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^E " gpurun_out/pytest_gpu.log | head -5
