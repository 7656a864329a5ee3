#!/bin/bash
# HIP stream priorities: the view streams above the input (augmentation) stream, and the reverse
for pair in "0 0" "-1 0" "0 -1" "0 0" "-1 0" "0 -1"; do
  set -- $pair
  SSV_VIEW_STREAM_PRIO=$1 SSV_INPUT_STREAM_PRIO=$2 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('view prio $1 input prio $2', 'images/s', d['value'], 'ms', d['ms_per_step'])"
done
