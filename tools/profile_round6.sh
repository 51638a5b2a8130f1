#!/bin/bash
# Round 6 profiles (runs on the MI355X box through gpurun; every step under its own timeout): rocprofv3 kernel-trace stats of
# the bench, of the one-pass resident matvec, of the end-to-end solve and of the per-config microbenchmarks; separate PMC
# passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) and the traffic*.json files bench.py cites; wave-cycle counters of the
# headline kernel and of the matvec; in-kernel phase timers (matvec, headline solve, pivoted Cholesky); the one-GPU
# emulations of the multi-rank gather; the bench line itself.  Output: gpurun_out/prof6/ (copied to profiles/r06/).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
MV="python $R/tools/mb_lowrank_mv.py 1 300"
stats() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/p_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- "$@" > $OUT/${name}_under_rocprof.log 2>&1
  local f=$(find /tmp/p_$name -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_$name.csv; fi
}
pmc() {  # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/q_${name}_$ctr
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/q_${name}_$ctr -- "$@" > /dev/null 2>&1
  timeout 60 python $R/tools/pmc_summary.py /tmp/q_${name}_$ctr
}
if [ "${1:-all}" != "late" ]; then
stats bench $B
stats mv $MV
stats mv_c2 python $R/tools/mb_lowrank_mv.py 2 300
stats mv_two_pass python $R/tools/mb_lowrank_mv.py 1 300 two-pass
stats e2e python $R/tools/mb_e2e.py
stats iql python $R/tools/mb_iql_pieces.py
{ pmc bench FETCH_SIZE $B; pmc bench WRITE_SIZE $B; } > $OUT/pmc_fetch_write_bench.txt
{ pmc mv FETCH_SIZE $MV; pmc mv WRITE_SIZE $MV; } > $OUT/pmc_fetch_write_mv.txt
{ pmc e2e FETCH_SIZE python $R/tools/mb_e2e.py; pmc e2e WRITE_SIZE python $R/tools/mb_e2e.py; } > $OUT/pmc_fetch_write_e2e.txt
{ pmc iql FETCH_SIZE python $R/tools/mb_iql_pieces.py; pmc iql WRITE_SIZE python $R/tools/mb_iql_pieces.py; } > $OUT/pmc_fetch_write_iql.txt
timeout 60 python - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
def grab(path, counter, kernel):
    try:
        for line in open(path):
            if line.startswith(counter) and kernel in line:
                return float(re.search(r"avg=\s*([0-9.]+)", line).group(1))
    except OSError:
        pass
    return None
for fname, src, prof_name, kern, label in (
        ("traffic.json", "pmc_fetch_write_bench.txt", "cg_onchip", "k_cg_rspace3<32, 8>", "k_cg_rspace3<32,8>"),
        ("traffic_mv.json", "pmc_fetch_write_mv.txt", "lr_mv", "k_lr_mv<32, 8, 1, 2>", "k_lr_mv<32,8,1>"),
        ("traffic_fused.json", "pmc_fetch_write_e2e.txt", "solve_fused", "k_solve_fused<32, 8, false>", "k_solve_fused<32,8,false>")):
    f, w = grab(f"{out}/{src}", "FETCH_SIZE", kern), grab(f"{out}/{src}", "WRITE_SIZE", kern)
    if f is not None and w is not None:
        json.dump({"prof_name": prof_name, "kernel": label, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w,
                   "fetch_correction": 2.0, "traffic_bytes_per_launch": (2.0 * f + w) * 1024,
                   "source": f"{src} (rocprofv3 --pmc, separate passes)"}, open(f"{out}/{fname}", "w"), indent=1)
rs = {}
for name, kern in (("rs_part", "k_rs_part"), ("rs_iter", "k_rs_iter"), ("rs_apply", "k_rs_apply"), ("pc_onchip", "k_pc_onchip4"),
                   ("pb_gram_root", "k_pb_gram_root"), ("rs_gram64", "k_rs_gram64")):
    f = grab(f"{out}/pmc_fetch_write_iql.txt", "FETCH_SIZE", kern)
    w = grab(f"{out}/pmc_fetch_write_iql.txt", "WRITE_SIZE", kern)
    if f is not None and w is not None:
        rs[name] = {"kernel": kern, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "fetch_correction": 2.0,
                    "traffic_bytes_per_launch": (2.0 * f + w) * 1024}
if rs:
    json.dump({"source": "pmc_fetch_write_iql.txt (rocprofv3 --pmc, separate passes, tools/mb_iql_pieces.py)",
               "kernels": rs}, open(f"{out}/traffic_iql.json", "w"), indent=1)
PY
: > $OUT/pmc_wave_cycles_headline.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS VALUBusy; do
  rm -rf /tmp/p_h
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_h -- $B > /dev/null 2>&1
  timeout 60 python $R/tools/pmc_summary.py /tmp/p_h k_cg_rspace3 >> $OUT/pmc_wave_cycles_headline.txt
done
: > $OUT/pmc_wave_cycles_mv.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS VALUBusy; do
  rm -rf /tmp/p_h
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_h -- $MV > /dev/null 2>&1
  timeout 60 python $R/tools/pmc_summary.py /tmp/p_h k_lr_mv >> $OUT/pmc_wave_cycles_mv.txt
done
fi
cd $R
# in-kernel phase timers
{ for m in 3 200 400; do LO_EIGFORM_AFTER_USES=0 LO_OC_DEBUG=$m timeout 100 python tools/mb_rspace_time.py 2>&1 | grep -E "onchip member" | tail -1; done; } > $OUT/rspace3_member_phases.txt
{ LO_MV_DEBUG=1 timeout 100 python tools/mb_lowrank_mv.py 1 3 2>&1 | grep -A8 -E "lr_mv group 0" | tail -9; } > $OUT/mv_member_phases.txt
{ echo "k_pc_onchip4<32,8,4>, 512 x 8192 x 32, rank 15: ONE group exchange per pivot = 15 per member (+ 1 placement check per launch and group)"; LO_OC_DEBUG=1 timeout 100 python tools/mb_pc_debug.py 2>&1 | grep -A1 "pc_onchip member0" | tail -4; } > $OUT/pivchol_member_phases.txt
# one-GPU emulations of the multi-rank gather: RCCL beside the solves, and the gather as peer writes
timeout 200 python tools/mb_dist_overlap.py > $OUT/dist_overlap.txt 2>&1
timeout 200 python tools/mb_peer_gather.py > $OUT/peer_gather.txt 2>&1
timeout 200 python tools/check_lowrank_mv.py --time > $OUT/check_lowrank_mv.txt 2>&1
# one training step (forward + backward through the operator API) as a kernel timeline with the idle gaps
cd /tmp; rm -rf /tmp/tl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/tools/iql_timeline.py train > /dev/null 2>&1
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then timeout 60 python $R/tools/iql_timeline.py parse $f > $OUT/iql_train_timeline.txt; fi
cd $R
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
