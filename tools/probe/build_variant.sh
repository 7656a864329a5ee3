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
# the variant's own build identity: "<shipped source hash>+<name>:<hash of the extra flags>" - a what-if library never answers ssv_source_sha16() like the shipped
# one, so bench.py's counters_stale check and the profile aggregators cannot mistake its counters for the product's
BASE=$(make -s -C $SRC print-src-sha)
FH=$(echo "$@" | sha256sum | cut -c1-8)
/opt/rocm/bin/hipcc $FLAGS "$@" -DSSV_SRC_SHA16=\"$BASE+$NAME:$FH\" -c $SRC/runtime.hip -o $OUT/runtime_$NAME.o
OTHERS=$(ls $SRC/*.o | grep -v -E "asan|conv_mfma.o|/bn.o|runtime.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/runtime_$NAME.o $OUT/conv_mfma_$NAME.o $OUT/bn_$NAME.o $OTHERS -o $OUT/libssv_$NAME.so
rm -f $OUT/conv_mfma_$NAME.o $OUT/bn_$NAME.o $OUT/runtime_$NAME.o
echo built $OUT/libssv_$NAME.so
