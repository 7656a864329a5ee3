#!/bin/bash
# r05 e7: the Winograd dispatch floors re-measured in the reference's CIFAR regime (resnet18 32x32, bs 512 / 64): step time through the step graph (GPU-bound
# there) with the channel floor at 64 (layer1's 64-channel 3x3 layers on Winograd) and the F(4x4) / F(2x2) tile floors lowered
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/e7_cifar_wino_floors.txt
: > $OUT
run() {  # label, env...
  label=$1; shift
  for bs in 512 64; do
    line=$(env "$@" python tools/bench_cifar.py $bs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['eager']['ms_per_step'], d['hip_graph']['ms_per_step'])")
    echo "$label bs $bs: eager / graph ms per step = $line" | tee -a $OUT
  done
}
run "default (channels >= 128, F(2x2) tiles >= 256, F(4x4) tiles >= 1024)" SSV_X=0
run "channel floor 64" SSV_WINOGRAD_MIN_CHANNELS=64
run "F(4x4) tile floor 256" SSV_WINOGRAD44_MIN_TILES=256
run "F(2x2) tile floor 64, F(4x4) tile floor 256" SSV_WINOGRAD_MIN_TILES=64 SSV_WINOGRAD44_MIN_TILES=256
run "channel floor 64 + F(4x4) tile floor 256" SSV_WINOGRAD_MIN_CHANNELS=64 SSV_WINOGRAD44_MIN_TILES=256
run "no Winograd" SSV_WINOGRAD=0
run "F(2x2) only" SSV_WINOGRAD44=0
