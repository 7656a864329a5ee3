#!/bin/bash
# Diagnostic library variants: tools/probe/build_variant.sh <name> <extra hipcc flags...>  ->  tools/probe/bin/libssv_<name>.so
# (conv_mfma.hip and bn.hip recompiled with the flags, every other object taken from the shipped build; select with SSV_HIP_LIB)
set -e
NAME=$1; shift
SRC=self-supervised-vision_amd/csrc
OUT=tools/probe/bin
mkdir -p $OUT
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-function"
for f in conv_mfma bn; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC/$f.hip -o $OUT/${f}_$NAME.o; done
OTHERS=$(ls $SRC/*.o | grep -v -E "asan|conv_mfma.o|/bn.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/conv_mfma_$NAME.o $OUT/bn_$NAME.o $OTHERS -o $OUT/libssv_$NAME.so
rm -f $OUT/conv_mfma_$NAME.o $OUT/bn_$NAME.o
echo built $OUT/libssv_$NAME.so
