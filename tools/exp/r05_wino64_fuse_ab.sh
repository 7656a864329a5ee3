#!/bin/bash
# r05 e11: 64-channel Winograd layers (now the default) with / without their input BatchNorm formed on load by the input transform; same box, 4 pairs
mkdir -p gpurun_out/r05
: > gpurun_out/r05/e11_wino64_fuse.txt
for i in 1 2 3 4; do
  for v in 0 1; do
    SSV_NO_NARROW_WINO_INPUT_FUSION=$v python bench.py --steps 15 --warmup 5 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e11_tmp.json 2> gpurun_out/r05/e11_tmp.err || { tail -20 gpurun_out/r05/e11_tmp.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/r05/e11_tmp.json')); print('pair $i no-fusion=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step', d['config']['peak_hbm_gb'], 'GB')" | tee -a gpurun_out/r05/e11_wino64_fuse.txt
  done
done
