#!/bin/bash
# kernel trace + phase timers of the one-pass resident matvec at the headline shape (every step under a timeout)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C=${1:-1}
rm -rf /tmp/p_mv
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mv -o mv -- python $R/tools/mb_lowrank_mv.py $C 300 > $OUT/under_rocprof_c$C.log 2>&1
grep "per batched" $OUT/under_rocprof_c$C.log
f=$(find /tmp/p_mv -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $OUT/kernel_stats_mv_c$C.csv; head -6 $f; fi
LO_MV_DEBUG=1 timeout 60 python $R/tools/mb_lowrank_mv.py $C 3 2>&1 | grep -E "lr_mv|per batched" | tail -3
for w in 1 2 3; do LO_MV_WGS_PER_CU=$w LO_MV_DEBUG=1 timeout 60 python $R/tools/mb_lowrank_mv.py $C 200 2>&1 | grep -E "lr_mv|per batched" | tail -2; done
