#!/bin/bash
python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -1
python tools/mb_cfg45.py cfg4 2>&1 | grep -E "CG:|cg_ctrl"
python tools/mb_cfg45.py cfg5 2>&1 | grep -E "CG:|cg_ctrl"
