#!/bin/bash
for v in 0 1; do for np in 0 1; do
  echo "== V2=$v NOPRE=$np"
  env $( [ $v = 1 ] && echo LO_LS_V2=1 ) $( [ $np = 1 ] && echo LS_NOPRE=1 ) python tools/mb_lockstep.py 2>&1 | grep -E "cg_solve|cg_lockstep"
done; done
