# static per-workgroup issue priority (arrival order on the CU mod 3 -> s_setprio 0 / 1 / 3) in the forward kernel: side library -DSSV_EXP_WGPRIO=1
export SSV_BENCH_LAYERS=stem,p64.0.conv2,p64.0.conv3,p256.0.conv3,p256.1.conv1,p128.0.conv3,p512.0.conv1
for v in shipped wgprio shipped wgprio; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/bench_conv.py 512 5 2>/dev/null | awk 'NR>2 && ($1=="stem" || $1 ~ /^p[0-9]/) && $1 !~ /:/ {printf "%-18s fwd %7s ms | dgrad %7s ms | wgrad %7s ms\n", $1, $10, $15, $19}'
done
