#!/bin/bash
python tools/mb_lockstep.py 2>&1 | grep -E "cg_solve|cg_lockstep|lockstep:"
python -m pytest tests/test_gpu_parity.py -q -x -k "lockstep" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do rm -rf /tmp/p_w; timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_w -- python $GRAFT_REPO_ROOT/tools/mb_lockstep.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p_w k_cg_lockstep; done
