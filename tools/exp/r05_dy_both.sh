#!/bin/bash
# r05 e5: one pass over dy for both backward operands of the Winograd F(4x4) layers + BatchNorm backward formed on load: tests, then same-box alternating A/B (5 pairs)
set -e
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_winograd44.py tests/test_gpu_r50_parity.py tests/test_gpu_trajectories.py -m gpu -x -q > gpurun_out/r05/e5_tests.log 2>&1 || { tail -40 gpurun_out/r05/e5_tests.log; exit 1; }
tail -2 gpurun_out/r05/e5_tests.log
: > gpurun_out/r05/e5_step_ab.txt
for i in 1 2 3 4 5; do
  for v in 1 0; do
    SSV_WINOGRAD44_DY_BOTH=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e5_tmp.json 2> gpurun_out/r05/e5_tmp.err || { tail -20 gpurun_out/r05/e5_tmp.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/r05/e5_tmp.json')); print('pair $i SSV_WINOGRAD44_DY_BOTH=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/r05/e5_step_ab.txt
  done
done
