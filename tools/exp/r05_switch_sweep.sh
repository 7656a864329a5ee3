#!/bin/bash
# r05 e13: every fusion of rounds 1-4 switched off one at a time on the round-5 library (headline step, one box, default before / between / after): does each still pay?
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/e13_switch_sweep.txt
: > $OUT
run() { label=$1; shift
  env "$@" python bench.py --steps 12 --warmup 4 --no-cpu-baseline --prof-steps 0 --no-other-configs > gpurun_out/r05/e13_tmp.json 2> gpurun_out/r05/e13_tmp.err || { echo "$label FAILED" | tee -a $OUT; tail -3 gpurun_out/r05/e13_tmp.err; return; }
  python -c "import json; d=json.load(open('gpurun_out/r05/e13_tmp.json')); print('%-36s' % '$label', d['ms_per_step'], 'ms/step', d['value'], 'images/s')" | tee -a $OUT
}
run default SSV_X=0
for sw in SSV_NO_BN_STATS_FUSION SSV_NO_BN_APPLY_FUSION SSV_NO_BN_APPLY_FUSION_3X3 SSV_NO_BN_BWD_FUSION SSV_NO_BN_DY_FUSION SSV_NO_CLOSING_FUSION; do run $sw=1 $sw=1; done
run default SSV_X=0
for sw in SSV_NO_SHORTCUT_GATE SSV_NO_COMPACT_S2_DGRAD SSV_NO_STEM_POOL_FUSION SSV_NO_POOLED_STEM_REDUCE SSV_LATE_LOSS_READ SSV_NO_INPUT_STREAM SSV_SINGLE_STREAM SSV_NO_NARROW_WINO_INPUT_FUSION; do run $sw=1 $sw=1; done
run "SSV_WINOGRAD44_DY_BOTH=0" SSV_WINOGRAD44_DY_BOTH=0
run default SSV_X=0
