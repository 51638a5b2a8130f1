#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -1
python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k: round(v['ms'],2) for k,v in d['other_configs'].items()})"
