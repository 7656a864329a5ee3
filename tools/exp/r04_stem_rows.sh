#!/bin/bash
# the stem's weight gradient from image rows staged in LDS (stem_wgrad_rows_k) against the row-taps gather (-DSSV_NO_STEM_ROWS side library)
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem" 2>&1 | tail -2
for v in shipped nostemrows; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  SSV_BENCH_LAYERS=stem python tools/bench_conv.py 512 10 gpurun_out/r04_stem.csv > /dev/null 2>&1
  echo "variant=$v $(sed -n 2p gpurun_out/r04_stem.csv | cut -d, -f1,9-11,18-20)"
done
