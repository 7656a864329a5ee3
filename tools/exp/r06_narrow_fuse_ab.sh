#!/bin/bash
# r06 e5: the 64-channel 3x3 layers, now on the direct bf16x3 kernels (128 x 64 tile), with their input BatchNorm + ReLU formed on load (shipped after this run; the experiment's switch SSV_EXP_FUSE_NARROW_3X3=1 became the default, off with SSV_NO_NARROW_3X3_INPUT_FUSION=1)
# instead of a materialised activation: images/s (two alternating rounds) and HBM traffic per step
set -u
export TMPDIR=/tmp
OUT=gpurun_out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 1"
PMCARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 0"
for rnd in 1 2 3; do
  for v in 0 1; do
    SSV_NO_NARROW_3X3_INPUT_FUSION=$((1 - v)) $B > $OUT/_ab.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("$OUT/_ab.json")); k = d["roofline"]["classes"]
print("fuse narrow 3x3 input = $v %8.2f images/s %8.3f ms/step  fwd %.1f dgrad %.1f wgrad %.1f bn %.1f ms " % (d["value"], d["ms_per_step"], k["conv_fwd"]["ms_per_step"], k["conv_dgrad"]["ms_per_step"],
      k["conv_wgrad"]["ms_per_step"], k["bn_fwd"]["ms_per_step"] + k["bn_bwd"]["ms_per_step"]))
PY
  done
done
for v in 0 1; do
  for n in fetch write; do
    rm -rf $OUT/_pmc_$n
    ctr=FETCH_SIZE; [ $n = write ] && ctr=WRITE_SIZE
    SSV_NO_NARROW_3X3_INPUT_FUSION=$((1 - v)) SSV_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/_pmc_$n -- python3 $PMCARGS > $OUT/_pmc_$n.log 2>&1
  done
  F=$(find $OUT/_pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $OUT/_pmc_write -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_traffic.py "$F" "$W" 2 $OUT/r06_e5_traffic_$v.json > /dev/null
  python3 - <<PY
import json
t = json.load(open("$OUT/r06_e5_traffic_$v.json"))["per_step_gb"]
tot = sum(x["fetch"] + x["write"] for x in t.values())
print("fuse = $v traffic %.1f GB/step: " % tot + ", ".join("%s %.1f" % (k, x["fetch"] + x["write"]) for k, x in sorted(t.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"]))[:6]))
PY
done
