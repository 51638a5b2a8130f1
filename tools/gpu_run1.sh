#!/bin/bash
python tools/mb_rank_sweep.py 2>&1 | grep "^rank"
python -m pytest tests -q -x -m gpu -k "precond or wide or sweep or above_32" 2>&1 | tail -2
