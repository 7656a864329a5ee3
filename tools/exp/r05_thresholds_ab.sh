#!/bin/bash
# r05 e12: the fusion thresholds of rounds 2-3 re-measured on the round-5 library (same box, alternating, 3 rounds): BatchNorm-backward operand formed on load from
# which map size (SSV_BN_DY_MIN_HW: default 784 = 28x28), closing activation formed by the next conv1 on which maps (SSV_CLOSING_HW: default 784,inf)
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/e12_thresholds.txt
: > $OUT
run() { label=$1; shift
  env "$@" python bench.py --steps 15 --warmup 5 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e12_tmp.json 2> gpurun_out/r05/e12_tmp.err || { tail -20 gpurun_out/r05/e12_tmp.err; return; }
  python -c "import json; d=json.load(open('gpurun_out/r05/e12_tmp.json')); print('$label', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
}
for i in 1 2 3; do
  run "round $i default" SSV_X=0
  run "round $i BN_DY_MIN_HW=196" SSV_BN_DY_MIN_HW=196
  run "round $i BN_DY_MIN_HW=49" SSV_BN_DY_MIN_HW=49
  run "round $i CLOSING_HW=196,inf" SSV_CLOSING_HW=196,1000000000
done
