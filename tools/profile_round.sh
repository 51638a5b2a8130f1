#!/bin/bash
# Runs on the MI355X box (gpurun): kernel-trace stats of the bench and of the end-to-end solve, then the two PMC
# passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) for the dominant kernel.  Output: gpurun_out/prof/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $B > $OUT/bench_under_rocprof.log 2>&1
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_e2e -- python $GRAFT_REPO_ROOT/tools/mb_e2e.py > $OUT/e2e_under_rocprof.log 2>&1
cp $(find /tmp/p_e2e -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_end_to_end.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cfg3 -- python $GRAFT_REPO_ROOT/tools/mb_cfg3.py > $OUT/cfg3_under_rocprof.log 2>&1
cp $(find /tmp/p_cfg3 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_cfg3_cg.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mr -- python $GRAFT_REPO_ROOT/tools/mb_minres.py > $OUT/minres_under_rocprof.log 2>&1
cp $(find /tmp/p_mr -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_minres.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- $B > /dev/null 2>&1
{ python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p_fetch; python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p_write; } > $OUT/pmc_fetch_write_summary.txt
cd $GRAFT_REPO_ROOT && timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.stderr
tail -c 600 $OUT/bench.json
