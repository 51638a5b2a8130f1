#!/bin/bash
LO_NO_RESIDENT_ORDER=1 timeout 300 python -m pytest tests/test_gpu_distributed.py -q -x -k two_host_threads 2>&1 | grep -E "passed|failed|assert|took|timed out" | head -8
