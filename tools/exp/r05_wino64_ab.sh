#!/bin/bash
# r05 e10: Winograd F(4x4) on the 64-channel 3x3 layers of layer1 (56x56) in the headline step?  SSV_WINOGRAD_MIN_CHANNELS=64 against the default 128, same box, 3 pairs
mkdir -p gpurun_out/r05
: > gpurun_out/r05/e10_wino64.txt
for i in 1 2 3; do
  for v in 64 128; do
    SSV_WINOGRAD_MIN_CHANNELS=$v python bench.py --steps 15 --warmup 5 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e10_tmp.json 2> gpurun_out/r05/e10_tmp.err || { tail -20 gpurun_out/r05/e10_tmp.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/r05/e10_tmp.json')); print('pair $i min channels $v', d['value'], 'images/s', d['ms_per_step'], 'ms/step', d['config']['peak_hbm_gb'], 'GB')" | tee -a gpurun_out/r05/e10_wino64.txt
  done
done
