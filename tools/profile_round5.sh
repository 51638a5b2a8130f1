#!/bin/bash
# Round 5 profiles (runs on the MI355X box through gpurun): rocprofv3 kernel-trace stats of the bench, of the end-to-end
# solve (fused one-launch kernel) and of the per-config microbenchmarks; separate PMC passes (FETCH_SIZE, WRITE_SIZE;
# kernel-trace only) for the headline kernel and the fused kernel, the traffic*.json files bench.py cites, wave-cycle
# counters of the fused kernel, and the bench line itself.  Output: gpurun_out/prof5/ (copied to profiles/r05/).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
stats() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/p_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- "$@" > $OUT/${name}_under_rocprof.log 2>&1
  cp "$(find /tmp/p_$name -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_$name.csv
}
stats bench $B
stats e2e python $R/tools/mb_e2e.py
stats iql python $R/tools/mb_iql_pieces.py
stats cfg45 python $R/tools/mb_cfg45.py
stats lanczos python $R/tools/mb_lanczos.py
# (fused_bign: unchanged since r04, see profiles/r04)
pmc() {  # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/q_${name}_$ctr
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/q_${name}_$ctr -- "$@" > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/q_${name}_$ctr
}
{ pmc bench FETCH_SIZE $B; pmc bench WRITE_SIZE $B; } > $OUT/pmc_fetch_write_bench.txt
{ LO_OC_NO_RSPACE=1 pmc bench3 FETCH_SIZE $B; LO_OC_NO_RSPACE=1 pmc bench3 WRITE_SIZE $B; } > $OUT/pmc_fetch_write_bench3.txt
{ pmc iql FETCH_SIZE python $R/tools/mb_iql_pieces.py; pmc iql WRITE_SIZE python $R/tools/mb_iql_pieces.py; } > $OUT/pmc_fetch_write_iql.txt
{ pmc e2e FETCH_SIZE python $R/tools/mb_e2e.py; pmc e2e WRITE_SIZE python $R/tools/mb_e2e.py; } > $OUT/pmc_fetch_write_e2e.txt
{ pmc cfg45 FETCH_SIZE python $R/tools/mb_cfg45.py; pmc cfg45 WRITE_SIZE python $R/tools/mb_cfg45.py; } > $OUT/pmc_fetch_write_cfg45.txt
{ pmc lockstep FETCH_SIZE python $R/tools/mb_lockstep.py; pmc lockstep WRITE_SIZE python $R/tools/mb_lockstep.py; } > $OUT/pmc_fetch_write_lockstep.txt
python - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
def grab(path, counter, kernel):
    for line in open(path):
        if line.startswith(counter) and kernel in line:
            return float(re.search(r"avg=\s*([0-9.]+)", line).group(1))
    return None
for fname, src, prof_name, kern, label in (
        ("traffic.json", "pmc_fetch_write_bench.txt", "cg_onchip", "k_cg_rspace<32, 8>", "k_cg_rspace<32,8>"),
        ("traffic_three_pass.json", "pmc_fetch_write_bench3.txt", "cg_onchip", "k_cg_onchip5<32, 8, 2>", "k_cg_onchip5<32,8,MODE 2 (w by recurrence)>"),
        ("traffic_fused.json", "pmc_fetch_write_e2e.txt", "solve_fused", "k_solve_fused<32, 8, false>", "k_solve_fused<32,8,false>")):
    f, w = grab(f"{out}/{src}", "FETCH_SIZE", kern), grab(f"{out}/{src}", "WRITE_SIZE", kern)
    if f is not None and w is not None:
        json.dump({"prof_name": prof_name, "kernel": label, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w,
                   "fetch_correction": 2.0, "traffic_bytes_per_launch": (2.0 * f + w) * 1024,
                   "source": f"{src} (rocprofv3 --pmc, separate passes)"}, open(f"{out}/{fname}", "w"), indent=1)
f, w = (grab(f"{out}/pmc_fetch_write_lockstep.txt", c, "k_cg_lockstep") for c in ("FETCH_SIZE", "WRITE_SIZE"))
if f is not None and w is not None:
    json.dump({"prof_name": "cg_lockstep", "kernel": "k_cg_lockstep<32,true,8>", "FETCH_SIZE_KB_avg": f,
               "WRITE_SIZE_KB_avg": w, "fetch_correction": 2.0, "traffic_bytes_per_launch": (2.0 * f + w) * 1024,
               "source": "pmc_fetch_write_lockstep.txt (rocprofv3 --pmc, separate passes, tools/mb_lockstep.py)"},
              open(f"{out}/traffic_lockstep.json", "w"), indent=1)
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
others = {}
for name, kern in (("kron_fused", "k_kron_fused"), ("precond_fused_kron", "k_precond_fused_kron"),
                   ("precond_fused", "k_precond_fused<"), ("dense_mv_mfma", "k_dense_mv_mfma16"), ("cg_step_cols", "k_cg_step_cols"),
                   ("kron_gemm_mfma", "k_kron_nt_mfma<true")):
    f = grab(f"{out}/pmc_fetch_write_cfg45.txt", "FETCH_SIZE", kern)
    w = grab(f"{out}/pmc_fetch_write_cfg45.txt", "WRITE_SIZE", kern)
    if f is not None and w is not None:
        others[name] = {"kernel": kern, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "fetch_correction": 2.0,
                        "traffic_bytes_per_launch": (2.0 * f + w) * 1024}
if others:
    json.dump({"source": "pmc_fetch_write_cfg45.txt (rocprofv3 --pmc, separate passes, tools/mb_cfg45.py)",
               "kernels": others}, open(f"{out}/traffic_cfg45.json", "w"), indent=1)
PY
: > $OUT/pmc_wave_cycles_headline.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS VALUBusy; do
  rm -rf /tmp/p_h
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_h -- $B > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/p_h k_cg_rspace >> $OUT/pmc_wave_cycles_headline.txt
done
: > $OUT/pmc_wave_cycles_headline_three_pass.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU VALUBusy; do
  rm -rf /tmp/p_h
  LO_OC_NO_RSPACE=1 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_h -- $B > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/p_h k_cg_onchip5 >> $OUT/pmc_wave_cycles_headline_three_pass.txt
done
# matrix-core utilisation of cfg3's multi-column R-space pass and of the lockstep kernel it replaced as first pass
: > $OUT/pmc_wave_cycles_iql.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  rm -rf /tmp/p_i
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_i -- python $R/tools/mb_iql_pieces.py > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/p_i k_rs_ >> $OUT/pmc_wave_cycles_iql.txt
done
# where the wave cycles of the fused kernel go (one pass per counter)
: > $OUT/pmc_wave_cycles_e2e.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 VALUBusy; do
  rm -rf /tmp/p_w
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_w -- python $R/tools/mb_e2e.py > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/p_w k_solve_fused >> $OUT/pmc_wave_cycles_e2e.txt
  python $R/tools/pmc_summary.py /tmp/p_w k_cg_onchip5 >> $OUT/pmc_wave_cycles_e2e.txt
done
cd $R
for m in 3 200 400; do LO_OC_DEBUG=$m python tools/mb_rspace_time.py 2>&1 | grep -E "onchip member" | tail -1; done > $OUT/rspace_member_phases.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 > $OUT/bench_cfg4_strong.json 2>> $OUT/bench.err
timeout 600 python bench.py --workload cfg5 --steps 1 --warmup 1 > $OUT/bench_cfg5_strong.json 2>> $OUT/bench.err
ls -la $OUT
