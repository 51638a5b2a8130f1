#!/bin/bash
# VALU / SALU / LDS utilisation of the dominant kernel (rocprofv3 derived metrics, one pass each, kernel-trace only).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
: > $OUT/pmc_utilisation.txt
for c in VALUBusy SALUBusy LdsUtil VALUUtilization SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES; do
  rm -rf /tmp/p_u
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_u -- $B > /dev/null 2>&1
  python - "$c" >> $OUT/pmc_utilisation.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
c = sys.argv[1]
rows = defaultdict(list)
for f in glob.glob("/tmp/p_u/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cg_onchip4" in r["Kernel_Name"] and r["Counter_Name"] == c:
            rows[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
if not rows:
    print(f"{c}: no rows"); sys.exit()
per = [sum(v) for v in rows.values()]
n_rows = len(next(iter(rows.values())))
print(f"{c}: dispatches {len(per)}, rows per dispatch {n_rows}, sum per dispatch avg {sum(per)/len(per):.4g}, mean row value {sum(per)/len(per)/n_rows:.4g}")
PY
done
cat $OUT/pmc_utilisation.txt
