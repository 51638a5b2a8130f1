#!/bin/bash
echo "== new swizzle (row & 7)"
for i in 1 2 3; do python tools/mb_lockstep.py 2>&1 | grep -E "cg_lockstep"; done
sed -i 's/((col >> 2) ^ (row \& 7))) << 2) + (col \& 3);/((col >> 2) ^ ((row >> 1) \& 7))) << 2) + (col \& 3);/' linear_operator_amd/csrc/lo_cg_lockstep.hip
grep -n "RC == 32) return" linear_operator_amd/csrc/lo_cg_lockstep.hip
make -C linear_operator_amd/csrc -j8 2>&1 | grep -E "error" | head -3
echo "== old swizzle ((row >> 1) & 7)"
for i in 1 2 3; do python tools/mb_lockstep.py 2>&1 | grep -E "cg_lockstep"; done
