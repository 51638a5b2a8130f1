#!/bin/bash
python tools/mb_bign_single.py 2>&1 | grep "^B="
python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -1
