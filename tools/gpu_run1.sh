#!/bin/bash
python tools/mb_dense_sweep.py 2>&1 | grep "^N="
