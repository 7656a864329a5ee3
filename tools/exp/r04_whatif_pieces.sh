# Per-piece what-if timings (diagnostic side libraries built with -DSSV_WHATIF=<bits>: 1 no MFMAs, 2 no epilogue stores, 4 activation loads through an empty descriptor)
# of the stem, the layer-1 3x3, the 64 -> 256 1x1 and an MFMA-bound 1x1: which piece is the time made of?   bash tools/exp/r04_whatif_pieces.sh
export SSV_BENCH_LAYERS=stem,p64.0.conv2,p64.0.conv3,p256.0.conv3
for v in shipped wi1 wi2 wi4 wi3 wi6; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/bench_conv.py 512 5 2>/dev/null | awk 'NR>2 && ($1=="stem" || $1 ~ /^p[0-9]/) {printf "%-18s fwd %-12s %7s ms | dgrad %-28s %7s ms | wgrad %-18s %7s ms\n", $1, $9, $10, $14, $15, $18, $19}'
done
