#!/bin/bash
# Evidence for one kernel state of the training step, written under gpurun_out/<tag>_*: run on the GPU box as
#   bash tools/profile_step.sh <tag> [bench.py arguments, e.g. --algo dino]
# 1. rocprofv3 --kernel-trace --stats of a single-stream run and of the two-stream run (per-kernel durations),
# 2. three separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES+GRBM_GUI_ACTIVE+wave stall buckets), each with
#    --kernel-trace only (counters never share a run with the other trace domains),
# 3. tools/pmc_traffic.py and tools/pmc_mfma.py aggregate them per kernel class and training step.
# The program after `--` is python3 itself (no wrapper that re-execs).
set -u
TAG=$1; shift
OUT=gpurun_out
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 0 $*"
PMCARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 0 $*"
stats() {   # $1 = name, env SSV_SINGLE_STREAM inherited
  rm -rf $OUT/_prof_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_prof_$1 -- python3 $ARGS > $OUT/_prof_$1.log 2>&1
  f=$(find $OUT/_prof_$1 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats_$1.csv
  # per-step sums from the trace itself: the --stats totals include the cold first step of the process (tools/kstats_steady.py)
  t=$(find $OUT/_prof_$1 -name '*kernel_trace.csv' | head -1)
  [ "$1" = single_stream ] && [ -n "$t" ] && python3 tools/kstats_steady.py "$t" > $OUT/${TAG}_family_time_per_step.txt
}
SSV_SINGLE_STREAM=1 stats single_stream
SSV_SINGLE_STREAM=0 stats two_streams
pmc() {     # $1 = name, rest = counters
  n=$1; shift
  rm -rf $OUT/_pmc_$n
  SSV_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/_pmc_$n -- python3 $PMCARGS > $OUT/_pmc_$n.log 2>&1
  find $OUT/_pmc_$n -name '*counter_collection.csv' | head -1
}
F=$(pmc fetch FETCH_SIZE)
W=$(pmc write WRITE_SIZE)
M=$(pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY)
MT=$(find $OUT/_pmc_mfma -name '*kernel_trace.csv' | head -1)
# 2 steps in each PMC run (1 warm-up + 1 timed)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_traffic.py "$F" "$W" 2 $OUT/${TAG}_pmc_hbm_traffic.json > /dev/null
[ -n "$M" ] && (cd tools && python3 pmc_mfma.py "../$M" 2 ../$OUT/${TAG}_pmc_mfma.json "${ALGO_GFLOP:--}" "../$MT" > /dev/null)
ls -la $OUT/${TAG}_* 2>/dev/null
tail -2 $OUT/_pmc_mfma.log
