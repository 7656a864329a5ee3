#!/bin/bash
# Evidence of round 6 on the current library, part 1 (profiles): rocprofv3 kernel stats (single / two streams), the three PMC passes with build identity, per-step family
# time, the per-variant counter table, the per-variant HBM traffic table and the where-the-time-goes table of the SimCLR bs-512 step; the 53-layer table.
# TAG defaults to r06 (files: profiles/r06_simclr_b512_*, r06_conv_layers_b512.csv)
export TMPDIR=/tmp
TAG=${1:-r06}
ALGO_GFLOP=24897.3 bash tools/profile_step.sh ${TAG}_simclr_b512 > gpurun_out/${TAG}_profile_simclr.log 2>&1
F=$(ls -t gpurun_out/_pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls -t gpurun_out/_pmc_write/*/*counter_collection.csv | head -1); T=$(ls -t gpurun_out/_prof_single_stream/*/*kernel_trace.csv | head -1)
python3 tools/pmc_traffic_variants.py $F $W $T 2 4 60 > gpurun_out/${TAG}_simclr_b512_traffic_by_variant.txt
bash tools/exp/r04_variants.sh ${TAG}_simclr_b512 > gpurun_out/${TAG}_variants_simclr.log 2>&1
python3 tools/where_time_goes.py gpurun_out/${TAG}_simclr_b512_kernel_variants.txt gpurun_out/${TAG}_simclr_b512_pmc_hbm_traffic.json > gpurun_out/${TAG}_simclr_b512_where_the_time_goes.txt
timeout -k 10 500 python tools/bench_conv.py 512 5 gpurun_out/${TAG}_conv_layers_b512.csv > gpurun_out/${TAG}_conv_layers_b512.txt 2>&1; tail -4 gpurun_out/${TAG}_conv_layers_b512.txt
for f in kernel_stats_single_stream.csv kernel_stats_two_streams.csv pmc_hbm_traffic.json pmc_mfma.json family_time_per_step.txt kernel_variants.txt traffic_by_variant.txt where_the_time_goes.txt; do cp gpurun_out/${TAG}_simclr_b512_$f profiles/ 2>/dev/null; done
cp gpurun_out/${TAG}_conv_layers_b512.csv profiles/
cat gpurun_out/${TAG}_simclr_b512_family_time_per_step.txt | tail -3
cat gpurun_out/${TAG}_simclr_b512_where_the_time_goes.txt
python3 -c "
import json
m=json.load(open('gpurun_out/${TAG}_simclr_b512_pmc_mfma.json')); t=json.load(open('gpurun_out/${TAG}_simclr_b512_pmc_hbm_traffic.json'))
print(m['summary']); print({k:(v['fetch']+v['write']) for k,v in t['per_step_gb'].items()}, sum(v['fetch']+v['write'] for v in t['per_step_gb'].values()))"
