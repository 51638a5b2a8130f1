#!/bin/bash
# Where the wave cycles of the two resident CG kernels go: one rocprofv3 --pmc pass per counter (kernel-trace only).
# Output: gpurun_out/prof2/pmc_wave_cycles_{bench,lockstep}.txt (sum over the XCD rows, average per dispatch).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE MemUnitStalled"
run() {  # name, kernel substring, command...
  local name=$1 kern=$2; shift 2
  : > $OUT/pmc_wave_cycles_$name.txt
  for c in $CTRS; do
    rm -rf /tmp/p_w
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_w -- "$@" > /dev/null 2>&1
    python - "$c" "$kern" >> $OUT/pmc_wave_cycles_$name.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
c, kern = sys.argv[1], sys.argv[2]
rows = defaultdict(float)
for f in glob.glob("/tmp/p_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == c:
            rows[r["Dispatch_Id"]] += float(r["Counter_Value"])
if not rows:
    print(f"{c}: no rows")
else:
    v = list(rows.values())
    print(f"{c}: {kern} dispatches {len(v)}, per dispatch avg {sum(v) / len(v):.6g}")
PY
  done
  cat $OUT/pmc_wave_cycles_$name.txt
}
run bench k_cg_onchip5 python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras
run lockstep k_cg_lockstep python $R/tools/mb_lockstep.py
