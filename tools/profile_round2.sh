#!/bin/bash
# Round 2 profiles (runs on the MI355X box through gpurun): rocprofv3 kernel-trace stats of the bench and of the
# per-config microbenchmarks, separate PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) for the headline kernel
# and the cfg3 lockstep kernel, the traffic.json bench.py cites, and the bench line itself.  Output: gpurun_out/prof2/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof2
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
stats cfg3 python $R/tools/mb_cfg3.py
stats lockstep python $R/tools/mb_lockstep.py
stats lanczos python $R/tools/mb_lanczos.py
stats cfg45 python $R/tools/mb_cfg45.py
stats iql python $R/tools/mb_iql_pieces.py
python $R/tools/mb_kron_rate.py > $OUT/kron_rate.txt 2>&1
pmc() {  # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/q_${name}_$ctr
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/q_${name}_$ctr -- "$@" > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/q_${name}_$ctr
}
{ pmc bench FETCH_SIZE $B; pmc bench WRITE_SIZE $B; } > $OUT/pmc_fetch_write_bench.txt
{ pmc cfg3 FETCH_SIZE python $R/tools/mb_lockstep.py; pmc cfg3 WRITE_SIZE python $R/tools/mb_lockstep.py; } > $OUT/pmc_fetch_write_lockstep.txt
{ pmc cfg45 FETCH_SIZE python $R/tools/mb_cfg45.py; pmc cfg45 WRITE_SIZE python $R/tools/mb_cfg45.py; } > $OUT/pmc_fetch_write_cfg45.txt
{ pmc lanczos FETCH_SIZE python $R/tools/mb_lanczos.py; pmc lanczos WRITE_SIZE python $R/tools/mb_lanczos.py; } > $OUT/pmc_fetch_write_lanczos.txt
# matrix-core / vector / LDS utilisation of the lockstep kernel (tools/pmc_lockstep.sh, same output directory)
bash $R/tools/pmc_lockstep.sh > /dev/null 2>&1
python - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
def grab(path, counter, kernel):
    for line in open(path):
        if line.startswith(counter) and kernel in line:
            return float(re.search(r"avg=\s*([0-9.]+)", line).group(1))
    return None
f = grab(f"{out}/pmc_fetch_write_bench.txt", "FETCH_SIZE", "k_cg_onchip5<32, 8, false>")
w = grab(f"{out}/pmc_fetch_write_bench.txt", "WRITE_SIZE", "k_cg_onchip5<32, 8, false>")
if f is not None and w is not None:
    json.dump({"prof_name": "cg_onchip", "kernel": "k_cg_onchip5<32,8,false>", "FETCH_SIZE_KB_avg": f,
               "WRITE_SIZE_KB_avg": w, "fetch_correction": 2.0, "traffic_bytes_per_launch": (2.0 * f + w) * 1024,
               "source": "pmc_fetch_write_bench.txt (rocprofv3 --pmc, separate passes, `bench.py --no-extras`)"},
              open(f"{out}/traffic.json", "w"), indent=1)
f = grab(f"{out}/pmc_fetch_write_lockstep.txt", "FETCH_SIZE", "k_cg_lockstep")
w = grab(f"{out}/pmc_fetch_write_lockstep.txt", "WRITE_SIZE", "k_cg_lockstep")
if f is not None and w is not None:
    json.dump({"prof_name": "cg_lockstep", "kernel": "k_cg_lockstep<32,true,8>", "FETCH_SIZE_KB_avg": f,
               "WRITE_SIZE_KB_avg": w, "fetch_correction": 2.0, "traffic_bytes_per_launch": (2.0 * f + w) * 1024,
               "source": "pmc_fetch_write_lockstep.txt (rocprofv3 --pmc, separate passes, tools/mb_lockstep.py)"},
              open(f"{out}/traffic_lockstep.json", "w"), indent=1)
others = {}
for name, kern in (("precond_fused", "k_precond_fused"), ("dense_mv_mfma", "k_dense_mv_mfma16"), ("kron_gemm_mfma", "k_kron_nt_mfma<true>"),
                   ("kron_gemm_mfma_first", "k_kron_nt_mfma<false>")):
    f = grab(f"{out}/pmc_fetch_write_cfg45.txt", "FETCH_SIZE", kern)
    w = grab(f"{out}/pmc_fetch_write_cfg45.txt", "WRITE_SIZE", kern)
    if f is not None and w is not None:
        others[name] = {"kernel": kern, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "fetch_correction": 2.0,
                        "traffic_bytes_per_launch": (2.0 * f + w) * 1024}
if others:
    json.dump({"source": "pmc_fetch_write_cfg45.txt (rocprofv3 --pmc, separate passes, tools/mb_cfg45.py)", "kernels": others},
              open(f"{out}/traffic_cfg45.json", "w"), indent=1)
PY
: > $OUT/pmc_utilisation_bench.txt
for c in VALUBusy SALUBusy LdsUtil VALUUtilization SQ_INSTS_VALU SQ_INSTS_LDS; do
  rm -rf /tmp/p_u
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_u -- $B > /dev/null 2>&1
  python - "$c" >> $OUT/pmc_utilisation_bench.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
c = sys.argv[1]
rows = defaultdict(list)
for f in glob.glob("/tmp/p_u/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cg_onchip5" in r["Kernel_Name"] and r["Counter_Name"] == c:
            rows[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
if not rows:
    print(f"{c}: no rows"); sys.exit()
per = [sum(v) for v in rows.values()]
print(f"{c}: k_cg_onchip5 dispatches {len(per)}, per dispatch avg {sum(per)/len(per):.6g}")
PY
done
cd $R && timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.stderr
tail -c 1500 $OUT/bench.json
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 > $OUT/bench_cfg4_strong.json 2>> $OUT/bench.stderr
timeout 900 python bench.py --workload cfg5 --steps 1 --warmup 0 --chunk-members 16 > $OUT/bench_cfg5_strong.json 2>> $OUT/bench.stderr
cat $OUT/bench_cfg4_strong.json $OUT/bench_cfg5_strong.json
tail -5 $OUT/bench.stderr
