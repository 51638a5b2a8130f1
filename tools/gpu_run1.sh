#!/bin/bash
python tools/mb_lowrank_cols.py 2>&1 | grep "^R="
