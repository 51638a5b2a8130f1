#!/bin/bash
# MFMA / VALU / LDS utilisation of the lockstep kernel (one rocprofv3 --pmc pass per counter, kernel-trace only)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/mb_lockstep.py"
: > $OUT/pmc_utilisation_lockstep.txt
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA VALUBusy MfmaUtil LdsUtil SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS; do
  rm -rf /tmp/p_u
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_u -- $CMD > /dev/null 2>&1
  python - "$c" >> $OUT/pmc_utilisation_lockstep.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
c = sys.argv[1]
rows = defaultdict(list)
for f in glob.glob("/tmp/p_u/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cg_lockstep" in r["Kernel_Name"] and r["Counter_Name"] == c:
            rows[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
if not rows:
    print(f"{c}: no rows"); sys.exit()
per = [sum(v) for v in rows.values()]
n_rows = len(next(iter(rows.values())))
print(f"{c}: dispatches {len(per)}, rows per dispatch {n_rows}, sum per dispatch avg {sum(per)/len(per):.6g}, mean row value {sum(per)/len(per)/n_rows:.6g}")
PY
done
cat $OUT/pmc_utilisation_lockstep.txt
