#!/bin/bash
# r06 e2: with fp32 products 1.4x cheaper (bf16x3) the Winograd layers' trade - 2.25-4x fewer multiplies for 2.25x the bytes of every operand, written and read back -
# has to be re-measured: the same step with the 64-channel layers, the 128-channel layers and every 3x3 layer on the direct kernels.
#   bash tools/exp/r06_wino_vs_direct_bf16x3.sh   (on the GPU box; writes gpurun_out/r06_e2_*.json)
set -u
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --prof-steps 1"
run() { name=$1; shift; env "$@" $B > gpurun_out/r06_e2_$name.json 2> gpurun_out/r06_e2_$name.err; python - <<PY
import json
d = json.load(open("gpurun_out/r06_e2_$name.json"))
c = d["roofline"]["classes"]
print("%-28s %8.2f images/s %8.3f ms/step  fwd %.1f dgrad %.1f wgrad %.1f bn %.1f ms" % ("$name", d["value"], d["ms_per_step"], c["conv_fwd"]["ms_per_step"], c["conv_dgrad"]["ms_per_step"],
      c["conv_wgrad"]["ms_per_step"], c["bn_fwd"]["ms_per_step"] + c["bn_bwd"]["ms_per_step"]))
PY
}
run shipped SSV_X=0
run wino_from_128 SSV_WINOGRAD44_MIN_CHANNELS=128
run wino_from_256 SSV_WINOGRAD44_MIN_CHANNELS=128 SSV_WINOGRAD_MIN_CHANNELS=256
run wino_from_512 SSV_WINOGRAD44_MIN_CHANNELS=128 SSV_WINOGRAD_MIN_CHANNELS=512
run no_winograd SSV_NO_WINOGRAD=1
run shipped_again SSV_X=0
