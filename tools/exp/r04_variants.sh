#!/bin/bash
# per kernel variant of the training step: time, clock, matrix-pipe busy, stall buckets (one counter pass, single stream; bench.py arguments after the tag)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_pmc_var
SSV_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/_pmc_var -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 0 "$@" > gpurun_out/_pmc_var.log 2>&1
python3 tools/pmc_variants.py $(find gpurun_out/_pmc_var -name '*counter_collection.csv' | head -1) $(find gpurun_out/_pmc_var -name '*kernel_trace.csv' | head -1) 2 45 | tee gpurun_out/${TAG}_kernel_variants.txt
rm -rf gpurun_out/_pmc_var
