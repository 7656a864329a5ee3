#!/bin/bash
# do the two view streams' kernels of the same kind collide?  view 1's forward / encoder backward started SSV_VIEW_SKEW_US / SSV_VIEW_SKEW_BWD_US behind view 0's
for pair in "0 0" "1500 0" "0 0" "1500 1500" "0 0" "800 3000" "0 0" "1500 5000"; do
  set -- $pair
  SSV_VIEW_SKEW_US=$1 SSV_VIEW_SKEW_BWD_US=$2 python bench.py --steps 25 --warmup 4 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skew_us fwd $1 bwd $2', 'images/s', d['value'], 'ms', d['ms_per_step'])"
done
