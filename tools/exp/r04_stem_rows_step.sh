#!/bin/bash
# the step with the stem's rows-in-LDS kernels (shipped) against the row-taps kernels (-DSSV_NO_STEM_ROWS side library), alternating on one box
for v in shipped nostemrows shipped nostemrows shipped nostemrows; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  python bench.py --steps 25 --warmup 4 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'images/s', d['value'], 'ms', d['ms_per_step'])"
done
