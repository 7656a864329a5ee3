#!/bin/bash
# counters of one layer's kernels in tools/bench_conv.py (SSV_BENCH_LAYERS=<substring>): matrix-pipe busy, stall buckets, effective clock.   r04_layer_pmc.sh <layer substring>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_layer
SSV_BENCH_LAYERS=$1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_layer -o run -- python3 tools/bench_conv.py 512 10 > gpurun_out/pmc_layer.log 2>&1
python3 tools/pmc_variants.py $(find gpurun_out/pmc_layer -name '*counter_collection.csv' | head -1) $(find gpurun_out/pmc_layer -name '*kernel_trace.csv' | head -1) 1 12
