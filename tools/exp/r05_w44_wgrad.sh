#!/bin/bash
# r05 e2: Winograd F(4x4) weight gradient - tests, per-layer probe (time + error vs fp64 at batch 512), same-box step A/B (5 alternating pairs)
set -e
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_winograd44.py tests/test_gpu_winograd.py -m gpu -x -q > gpurun_out/r05/e2_tests.log 2>&1 || { tail -40 gpurun_out/r05/e2_tests.log; exit 1; }
tail -3 gpurun_out/r05/e2_tests.log
python tools/probe_winograd44_wgrad.py > gpurun_out/r05/e2_probe_winograd44_wgrad.txt 2>&1 || { tail -30 gpurun_out/r05/e2_probe_winograd44_wgrad.txt; exit 1; }
cat gpurun_out/r05/e2_probe_winograd44_wgrad.txt
: > gpurun_out/r05/e2_step_ab.txt
for i in 1 2 3 4 5; do
  for v in 1 0; do
    SSV_WINOGRAD44_FWD_RATIO=${FWD_RATIO:-0.75} SSV_WINOGRAD44_WGRAD=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e2_tmp.json 2> gpurun_out/r05/e2_tmp.err || { tail -20 gpurun_out/r05/e2_tmp.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/r05/e2_tmp.json')); print('pair $i SSV_WINOGRAD44_WGRAD=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/r05/e2_step_ab.txt
  done
done
