#!/bin/bash
# r06: generic same-box A/B of environment switches on the headline step: images/s (alternating rounds) and HBM traffic per step (FETCH_SIZE / WRITE_SIZE passes).
#   [BENCH_EXTRA="--algo dino"] bash tools/exp/r06_env_ab.sh <rounds> "name1:ENV=V ENV2=V" "name2:..." ...      (a first configuration "shipped:SSV_X=0" is always added)
set -u
export TMPDIR=/tmp
OUT=gpurun_out
ROUNDS=$1; shift
B="python bench.py ${BENCH_EXTRA:-} --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 1"
PMCARGS="bench.py ${BENCH_EXTRA:-} --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 0"
CFGS=("shipped:SSV_X=0" "$@")
for rnd in $(seq 1 $ROUNDS); do
  for c in "${CFGS[@]}"; do
    name=${c%%:*}; envs=${c#*:}
    env $envs $B > $OUT/_ab.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("$OUT/_ab.json")); k = d["roofline"]["classes"]
ms = lambda c: k.get(c, {}).get("ms_per_step", 0.0)
print("%-14s %8.2f images/s %8.3f ms/step  fwd %.1f dgrad %.1f wgrad %.1f bn %.1f attn %.1f norm %.1f ms" % ("$name", d["value"], d["ms_per_step"], ms("conv_fwd"), ms("conv_dgrad"),
      ms("conv_wgrad"), ms("bn_fwd") + ms("bn_bwd"), ms("attn"), ms("norm")))
PY
  done
done
for c in "${CFGS[@]}"; do
  name=${c%%:*}; envs=${c#*:}
  for n in fetch write; do
    rm -rf $OUT/_pmc_$n
    ctr=FETCH_SIZE; [ $n = write ] && ctr=WRITE_SIZE
    env $envs SSV_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/_pmc_$n -- python3 $PMCARGS > $OUT/_pmc_$n.log 2>&1
  done
  F=$(find $OUT/_pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $OUT/_pmc_write -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_traffic.py "$F" "$W" 2 $OUT/_ab_traffic.json > /dev/null
  python3 - <<PY
import json
t = json.load(open("$OUT/_ab_traffic.json"))["per_step_gb"]
tot = sum(x["fetch"] + x["write"] for x in t.values())
print("%-14s traffic %.1f GB/step: " % ("$name", tot) + ", ".join("%s %.1f" % (k, x["fetch"] + x["write"]) for k, x in sorted(t.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"]))[:6]))
PY
done
