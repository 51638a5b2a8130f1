#!/bin/bash
# headline kernel with the groups started in P phases, D us apart (LO_OC_STAGGER / LO_OC_STAGGER_US)
run() { python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.readlines()[-1]); print('$1', b['ms_per_step'], b['roofline']['avg_launch_us'], b['final_mean_residual'])"; }
run base
for P in 2 4 8; do for D in 6 12 25; do export LO_OC_STAGGER=$P LO_OC_STAGGER_US=$D; run "P=$P D=$D"; done; done
