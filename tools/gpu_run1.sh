#!/bin/bash
python -m pytest tests/test_gpu_distributed.py -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
for r in 0 32; do echo "== reserve $r"; LO_OC_RESERVE_CUS=$r python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
LO_OC_RESERVE_CUS=32 python tools/mb_iql_pieces.py 2>&1 | grep -E "^iql|cg_lockstep|cg_onchip|pc_onchip"
