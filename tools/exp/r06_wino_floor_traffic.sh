#!/bin/bash
# r06 e4: bytes against multiplies on the final library - the step with the Winograd channel floor at 64 (shipped), 128, 256 and 512 channels: images/s (two alternating
# rounds) and HBM traffic per step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.py).  Review item 2 asks for <= 620 GB per step.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 1"
PMCARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 0"
cfg() { case $1 in
  shipped) echo "SSV_X=0";;
  from128) echo "SSV_WINOGRAD44_MIN_CHANNELS=128";;
  from256) echo "SSV_WINOGRAD44_MIN_CHANNELS=256 SSV_WINOGRAD_MIN_CHANNELS=256";;
  from512) echo "SSV_WINOGRAD44_MIN_CHANNELS=512 SSV_WINOGRAD_MIN_CHANNELS=512";;
esac; }
for rnd in 1 2; do
  for c in shipped from128 from256 from512; do
    env $(cfg $c) $B > $OUT/_ab.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("$OUT/_ab.json")); k = d["roofline"]["classes"]
print("%-8s %8.2f images/s %8.3f ms/step  fwd %.1f dgrad %.1f wgrad %.1f bn %.1f ms" % ("$c", d["value"], d["ms_per_step"], k["conv_fwd"]["ms_per_step"], k["conv_dgrad"]["ms_per_step"],
      k["conv_wgrad"]["ms_per_step"], k["bn_fwd"]["ms_per_step"] + k["bn_bwd"]["ms_per_step"]))
PY
  done
done
for c in shipped from128 from256 from512; do
  for n in fetch write; do
    rm -rf $OUT/_pmc_$n
    ctr=FETCH_SIZE; [ $n = write ] && ctr=WRITE_SIZE
    env $(cfg $c) SSV_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/_pmc_$n -- python3 $PMCARGS > $OUT/_pmc_$n.log 2>&1
  done
  F=$(find $OUT/_pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $OUT/_pmc_write -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_traffic.py "$F" "$W" 2 $OUT/r06_e4_traffic_$c.json > /dev/null
  python3 - <<PY
import json
t = json.load(open("$OUT/r06_e4_traffic_$c.json"))["per_step_gb"]
tot = sum(v["fetch"] + v["write"] for v in t.values())
print("%-8s traffic %.1f GB/step: " % ("$c", tot) + ", ".join("%s %.1f" % (k, v["fetch"] + v["write"]) for k, v in sorted(t.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"]))[:6]))
PY
done
