#!/bin/bash
O=gpurun_out/prof2; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.stderr; tail -c 600 $O/bench.json
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 > $O/bench_cfg4_strong.json 2>> $O/bench.stderr
timeout 900 python bench.py --workload cfg5 --steps 1 --warmup 0 --chunk-members 16 > $O/bench_cfg5_strong.json 2>> $O/bench.stderr
python -c "
import json
for f in ('bench','bench_cfg4_strong','bench_cfg5_strong'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
for r in d['rooflines']: print(r['kernel'], r['bound'], round(r['frac'],3), round(r['avg_launch_us'],1), r.get('traffic'))
for k,v in d['other_configs'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
print(d['cpu_baseline'])
"
