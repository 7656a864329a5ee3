#!/bin/bash
# r05 e9: does replaying the HEADLINE step (ResNet-50 224x224 bs 512) as one HIP graph buy anything?  same-box alternating A/B, 3 pairs
mkdir -p gpurun_out/r05
: > gpurun_out/r05/e9_graph_224.txt
for i in 1 2 3; do
  for v in 1 0; do
    SSV_BENCH_VIA_STEP=$v SSV_STEP_GRAPH=$v python bench.py --steps 15 --warmup 6 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e9_tmp.json 2> gpurun_out/r05/e9_tmp.err || { tail -20 gpurun_out/r05/e9_tmp.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/r05/e9_tmp.json')); print('pair $i graph=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step', d['config']['peak_hbm_gb'], 'GB')" | tee -a gpurun_out/r05/e9_graph_224.txt
  done
done
