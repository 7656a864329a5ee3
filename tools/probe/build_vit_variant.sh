#!/bin/bash
# Diagnostic variants of the attention kernels: tools/probe/build_vit_variant.sh <name> <patch file or -> <extra hipcc flags...>  ->  tools/probe/bin/libssv_attn_<name>.so
# (csrc/vit.hip - with the patch applied to a scratch copy, the tree stays as it is - recompiled with the flags, every other object from the shipped build; its own build
# identity "<shipped source hash>+attn_<name>:<hash of patch and flags>", so no counter file of the product can be mistaken for it; select with SSV_HIP_LIB).
# The r05 attention experiments (profiles/r05_probe_attention.txt):
#   base                         -                                          (= the shipped kernel, for alternating runs)
#   w0 w1 w2 w4 w8 w15 w16 w31   tools/exp/r05_attn_whatif.patch  -DSSV_ATTN_WHATIF=<n>
#   p3 p2 late3 late2            tools/exp/r05_attn_persistent_forward.patch  [-DSSV_ATTN_FWD_OCC=2] [-DSSV_ATTN_QNEXT_LATE]
#   grid                         tools/exp/r05_attn_grid.patch
#   stagger                      tools/exp/r05_attn_stagger.patch   (run with SSV_ATTN_STAGGER=<n>)
#   bwd_new / bwd_old            tools/exp/r05_attn_bwd_loads.patch  [-DSSV_ATTN_BWD_LOADS_AT_TOP]
set -e
NAME=$1; PATCH=$2; shift 2
SRC=self-supervised-vision_amd/csrc
OUT=tools/probe/bin
mkdir -p $OUT
TMP=$(mktemp -d)
cp $SRC/vit.hip $TMP/
if [ "$PATCH" != "-" ]; then (cd $TMP && patch -s -p3 < $OLDPWD/$PATCH); fi
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-function -I$SRC -I$PWD/$SRC"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $TMP/vit.hip -o $OUT/vit_$NAME.o
BASE=$(make -s -C $SRC print-src-sha)
FH=$( (cat $TMP/vit.hip; echo "$@") | sha256sum | cut -c1-8)
# runtime.hip includes every source for its hash at the shipped path; only its identity string changes here
/opt/rocm/bin/hipcc $FLAGS -DSSV_SRC_SHA16=\"$BASE+attn_$NAME:$FH\" -c $SRC/runtime.hip -o $OUT/runtime_$NAME.o
OTHERS=$(ls $SRC/*.o | grep -v -E "asan|/vit.o|/runtime.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/runtime_$NAME.o $OUT/vit_$NAME.o $OTHERS -o $OUT/libssv_attn_$NAME.so
rm -rf $OUT/vit_$NAME.o $OUT/runtime_$NAME.o $TMP
echo built $OUT/libssv_attn_$NAME.so
