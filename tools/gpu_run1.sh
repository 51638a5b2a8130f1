#!/bin/bash
python tools/mb_rootform.py 2>&1 | grep "^B="
python -m pytest tests/test_gpu_parity.py -q -x -k "root_form or onchip or lockstep or precond" 2>&1 | tail -2
