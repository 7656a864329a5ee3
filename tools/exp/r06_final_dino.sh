#!/bin/bash
# Evidence of round 6, part 2: the DINO ViT-S/16 bs-128 step's profiles (kernel stats, PMC passes, variants)
export TMPDIR=/tmp
TAG=${1:-r06}
bash tools/profile_step.sh ${TAG}_dino_b128 --algo dino > gpurun_out/${TAG}_profile_dino.log 2>&1
bash tools/exp/r04_variants.sh ${TAG}_dino_b128 --algo dino > gpurun_out/${TAG}_variants_dino.log 2>&1
python3 tools/where_time_goes.py gpurun_out/${TAG}_dino_b128_kernel_variants.txt gpurun_out/${TAG}_dino_b128_pmc_hbm_traffic.json > gpurun_out/${TAG}_dino_b128_where_the_time_goes.txt
for f in kernel_stats_single_stream.csv kernel_stats_two_streams.csv pmc_hbm_traffic.json pmc_mfma.json family_time_per_step.txt kernel_variants.txt where_the_time_goes.txt; do cp gpurun_out/${TAG}_dino_b128_$f profiles/ 2>/dev/null; done
tail -3 gpurun_out/${TAG}_dino_b128_family_time_per_step.txt; cat gpurun_out/${TAG}_dino_b128_where_the_time_goes.txt
