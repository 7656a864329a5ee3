#!/bin/bash
# r05 e8: the stride-2 3x3 trio (p128.0 / p256.0 / p512.0 conv2: 18.9 ms/step at 104-118 TFLOP/s) piece by piece - what-if side libraries (-DSSV_WHATIF=<bits>: 1 no MFMAs,
# 2 no epilogue stores, 4 activation loads through an empty descriptor), then each layer's own counters (matrix-pipe busy, stall buckets, clock).
# Build first, in the build container:  for v in 1 2 4 3 6; do bash tools/probe/build_variant.sh wi$v -DSSV_WHATIF=$v; done
export SSV_BENCH_LAYERS=p128.0.conv2,p256.0.conv2,p512.0.conv2
OUT=gpurun_out/r05/e8_whatif_s2.txt
mkdir -p gpurun_out/r05
: > $OUT
for v in shipped wi1 wi2 wi4 wi3 wi6; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  echo "== $v" | tee -a $OUT
  timeout -k 10 200 python tools/bench_conv.py 512 5 2>/dev/null | awk 'NR>2 && ($1 ~ /^p[0-9]/) {printf "%-18s fwd %-12s %7s ms | dgrad %-28s %7s ms | wgrad %-18s %7s ms\n", $1, $9, $10, $14, $15, $18, $19}' | tee -a $OUT
done
unset SSV_HIP_LIB
for l in p128.0.conv2 p256.0.conv2 p512.0.conv2; do
  echo "== counters $l" | tee -a $OUT
  bash tools/exp/r04_layer_pmc.sh $l 2>&1 | tail -8 | cut -c1-170 | tee -a $OUT
done
