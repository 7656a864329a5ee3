#!/bin/bash
# r06 e3: the formed-on-load thresholds (which feature-map sizes let the 1x1 consumers form the BatchNorm backward / the closing activation while they stage it) were
# measured on the fp32 MFMA instruction, where the small deep maps' consumers were matrix-bound.  Re-measured in the bf16x3 arithmetic.
#   bash tools/exp/r06_thresholds_bf16x3.sh   (on the GPU box; writes gpurun_out/r06_e3_*.json, prints one line per state)
set -u
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 1"
run() { name=$1; shift; env "$@" $B > gpurun_out/r06_e3_$name.json 2> gpurun_out/r06_e3_$name.err; python - <<PY
import json
d = json.load(open("gpurun_out/r06_e3_$name.json"))
c = d["roofline"]["classes"]
print("%-28s %8.2f images/s %8.3f ms/step  fwd %.1f dgrad %.1f wgrad %.1f bn %.1f ms" % ("$name", d["value"], d["ms_per_step"], c["conv_fwd"]["ms_per_step"], c["conv_dgrad"]["ms_per_step"],
      c["conv_wgrad"]["ms_per_step"], c["bn_fwd"]["ms_per_step"] + c["bn_bwd"]["ms_per_step"]))
PY
}
run shipped SSV_X=0
run bn_dy_from_196 SSV_BN_DY_MIN_HW=196
run bn_dy_all SSV_BN_DY_MIN_HW=0
run closing_from_196 SSV_CLOSING_HW=196,1000000000
run closing_all SSV_CLOSING_HW=0,1000000000
run both_from_196 SSV_BN_DY_MIN_HW=196 SSV_CLOSING_HW=196,1000000000
run both_all SSV_BN_DY_MIN_HW=0 SSV_CLOSING_HW=0,1000000000
run shipped_again SSV_X=0
