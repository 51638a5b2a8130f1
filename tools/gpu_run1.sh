#!/bin/bash
python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -1
LO_CG_GRAPH=1 python -m pytest tests -q -x -m gpu -k "kron or dense or fullsize or sweep or fused" 2>&1 | grep -E "passed|failed" | tail -1
python tools/mb_cfg45.py cfg4 2>&1 | grep -E "CG:"
