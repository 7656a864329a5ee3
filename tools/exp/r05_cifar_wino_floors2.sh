#!/bin/bash
# r05 e7b: more floor combinations (see r05_cifar_wino_floors.sh)
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/e7b_cifar_wino_floors.txt
: > $OUT
run() {  # label, env...
  label=$1; shift
  for bs in 512 64; do
    line=$(env "$@" python tools/bench_cifar.py $bs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['eager']['ms_per_step'], d['hip_graph']['ms_per_step'])")
    echo "$label bs $bs: eager / graph ms per step = $line" | tee -a $OUT
  done
}
run "default" SSV_X=0
run "no Winograd" SSV_NO_WINOGRAD=1
run "F(2x2) floor 64, F(4x4) floor 256" SSV_WINOGRAD_MIN_TILES=64 SSV_WINOGRAD44_MIN_TILES=256
run "F(2x2) floor 16, F(4x4) floor 64" SSV_WINOGRAD_MIN_TILES=16 SSV_WINOGRAD44_MIN_TILES=64
run "F(2x2) floor 16, F(4x4) floor 256" SSV_WINOGRAD_MIN_TILES=16 SSV_WINOGRAD44_MIN_TILES=256
run "F(2x2) floor 64, F(4x4) floor 1024" SSV_WINOGRAD_MIN_TILES=64
run "channels 64, F(2x2) floor 64, F(4x4) floor 256" SSV_WINOGRAD_MIN_CHANNELS=64 SSV_WINOGRAD_MIN_TILES=64 SSV_WINOGRAD44_MIN_TILES=256
