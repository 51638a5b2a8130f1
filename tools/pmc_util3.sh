#!/bin/bash
# Round 3: matrix-core / VALU / LDS utilisation of the lockstep kernel and of the two cfg4 kernels (one rocprofv3 --pmc
# pass per counter, kernel-trace only).  Output: gpurun_out/prof3/pmc_utilisation_{lockstep,cfg4}.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # out file, kernel substrings (comma separated), command...
  local out=$1 kerns=$2; shift 2
  : > $out
  for c in MfmaUtil VALUBusy LdsUtil SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; do
    rm -rf /tmp/p_u
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_u -- "$@" > /dev/null 2>&1
    python - "$c" "$kerns" >> $out <<'PY'
import csv, glob, sys
from collections import defaultdict
c, kerns = sys.argv[1], sys.argv[2].split(",")
for kern in kerns:
    rows = defaultdict(list)
    for f in glob.glob("/tmp/p_u/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == c:
                rows[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    if not rows:
        print(f"{c:28s} {kern}: no rows"); continue
    per = [sum(v) for v in rows.values()]
    n_rows = len(next(iter(rows.values())))
    print(f"{c:28s} {kern}: dispatches {len(per)}, rows per dispatch {n_rows}, sum per dispatch avg {sum(per)/len(per):.6g}, mean row value {sum(per)/len(per)/n_rows:.6g}")
PY
  done
}
run $OUT/pmc_utilisation_lockstep.txt k_cg_lockstep python $GRAFT_REPO_ROOT/tools/mb_lockstep.py
run $OUT/pmc_utilisation_cfg4.txt k_kron_fused,k_precond_fused_kron python $GRAFT_REPO_ROOT/tools/mb_cfg45.py cfg4
cat $OUT/pmc_utilisation_lockstep.txt $OUT/pmc_utilisation_cfg4.txt
