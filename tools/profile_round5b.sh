#!/bin/bash
# Round 5, second half (the diagonal form of the R-space iteration): the headline pieces of tools/profile_round5.sh on
# the new build -- kernel-trace stats of the bench, FETCH_SIZE / WRITE_SIZE passes (traffic.json), wave-cycle counters of
# k_cg_rspace<32,8,true>, LO_OC_DEBUG member phases (diagonal and dense form), the eigen-form build, the bench line.
# Output: gpurun_out/prof5b/ (copied to profiles/r05/).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof5b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
rm -rf /tmp/p_bench
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- $B > $OUT/bench_under_rocprof.log 2>&1
cp "$(find /tmp/p_bench -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_bench.csv
pmc() {  # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/q_${name}_$ctr
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/q_${name}_$ctr -- "$@" > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/q_${name}_$ctr
}
{ pmc bench FETCH_SIZE $B; pmc bench WRITE_SIZE $B; } > $OUT/pmc_fetch_write_bench.txt
python - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
def grab(path, counter, kernel):
    for line in open(path):
        if line.startswith(counter) and kernel in line:
            return float(re.search(r"avg=\s*([0-9.]+)", line).group(1))
    return None
for fname, kern, label in (("traffic.json", "k_cg_rspace<32, 8, true>", "k_cg_rspace<32,8,true>"),
                           ("traffic_rspace_dense.json", "k_cg_rspace<32, 8, false>", "k_cg_rspace<32,8,false>")):
    f, w = grab(f"{out}/pmc_fetch_write_bench.txt", "FETCH_SIZE", kern), grab(f"{out}/pmc_fetch_write_bench.txt", "WRITE_SIZE", kern)
    if f is not None and w is not None:
        json.dump({"prof_name": "cg_onchip", "kernel": label, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w,
                   "fetch_correction": 2.0, "traffic_bytes_per_launch": (2.0 * f + w) * 1024,
                   "source": "pmc_fetch_write_bench.txt (rocprofv3 --pmc, separate passes)"}, open(f"{out}/{fname}", "w"), indent=1)
PY
: > $OUT/pmc_wave_cycles_headline.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS VALUBusy; do
  rm -rf /tmp/p_h
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_h -- $B > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/p_h k_cg_rspace >> $OUT/pmc_wave_cycles_headline.txt
done
cd $R/tools
python mb_rsdiag_phases.py 2>&1 | grep -E "^---|onchip member" > $OUT/rspace_member_phases.txt
rm -rf /tmp/p_e
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_e -- python check_eigform.py > $OUT/check_eigform.txt 2>&1
grep -E "Name|eigform|k_cg_rspace|k_rs_gram64|k_pb_rootform|k_pc_onchip4" "$(find /tmp/p_e -name '*kernel_stats.csv' | head -1)" | cut -c1-220 > $OUT/kernel_stats_eigform.csv
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
