# Winograd F(4x4) on / off (SSV_WINOGRAD44=0 = F(2x2) everywhere, round 3's arithmetic) on the three ResNet-50 workloads, one box
for i in 1 2; do
  for v in f44 f22; do
    if [ $v = f22 ]; then export SSV_WINOGRAD44=0; else unset SSV_WINOGRAD44; fi
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('simclr_$v_$i'.replace('_$i','') + '_$i', d['value'], d['ms_per_step'])"
  done
done
for a in byol barlow; do
  for v in f44 f22; do
    if [ $v = f22 ]; then export SSV_WINOGRAD44=0; else unset SSV_WINOGRAD44; fi
    timeout -k 10 300 python bench.py --algo $a --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${a}_$v', d['value'], d['ms_per_step'])"
  done
done
unset SSV_WINOGRAD44
