cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "root_form or onchip or lockstep" 2>&1 | tail -8
LO_OC_DEBUG=5 LS_C=1 timeout 300 python tools/mb_lockstep.py 2>&1 | grep -v amdgpu.ids | tail -4
LS_C=1 timeout 300 python tools/mb_lockstep.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print('value', o['value'], 'ms', o['ms_per_step'], 'e2e', o['end_to_end_ms'], 'roof', o['roofline']['frac'], o['roofline']['avg_launch_us'])"
