#!/bin/bash
python tools/mb_kron_batch.py 2>&1 | grep "^n="
