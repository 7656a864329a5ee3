#!/bin/bash
# r05 e3: with the F(4x4) weight gradient shipped, does the 14x14 FORWARD switch to F(4x4) too?  same-box alternating A/B, 5 pairs (0.75 = 14x14 on F(4x4), 0.6 = F(2x2))
set -e
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_winograd44.py -m gpu -x -q > gpurun_out/r05/e3_tests.log 2>&1 || { tail -40 gpurun_out/r05/e3_tests.log; exit 1; }
tail -2 gpurun_out/r05/e3_tests.log
: > gpurun_out/r05/e3_step_ab.txt
for i in 1 2 3 4 5; do
  for v in 0.75 0.6; do
    SSV_WINOGRAD44_FWD_RATIO=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e3_tmp.json 2> gpurun_out/r05/e3_tmp.err || { tail -20 gpurun_out/r05/e3_tmp.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/r05/e3_tmp.json')); print('pair $i SSV_WINOGRAD44_FWD_RATIO=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/r05/e3_step_ab.txt
  done
done
