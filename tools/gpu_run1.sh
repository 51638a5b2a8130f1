#!/bin/bash
python tools/mb_host_batched.py 2>&1 | grep "^B="
echo "== groups of eight"
LO_OC_GW8=1 python tools/mb_host_batched.py 2>&1 | grep "^B=" | head -3
