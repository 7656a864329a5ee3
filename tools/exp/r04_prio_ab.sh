# s_setprio experiment: waves raise their issue priority while they are in the MFMA part of a k-tile (side libraries built with -DSSV_EXP_PRIO=1|3)
export SSV_BENCH_LAYERS=stem,p64.0.conv2,p64.0.conv3,p256.0.conv3,p256.1.conv1,p128.0.conv3
for v in shipped prio1 prio3; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/bench_conv.py 512 5 2>/dev/null | awk 'NR>2 && ($1=="stem" || $1 ~ /^p[0-9]/) {printf "%-18s fwd %7s ms | dgrad %7s ms | wgrad %7s ms\n", $1, $10, $15, $19}'
done
unset SSV_BENCH_LAYERS
for i in 1 2; do
  for v in shipped prio1 prio3; do
    if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('simclr $v $i', d['value'], d['ms_per_step'])"
  done
done
