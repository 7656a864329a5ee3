#!/bin/bash
# s_memtime stamps of the one-pass attention backward (diagnostic build of vit.hip with -DSSV_STAMP_ATTN, other objects from the shipped build)
set -e
SRC=self-supervised-vision_amd/csrc; OUT=tools/probe/bin; mkdir -p $OUT
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-function -DSSV_STAMP_ATTN -c $SRC/vit.hip -o $OUT/vit_stamp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/vit_stamp.o $(ls $SRC/*.o | grep -v -E "asan|/vit.o") -o $OUT/libssv_attnstamp.so
  rm -f $OUT/vit_stamp.o; echo built $OUT/libssv_attnstamp.so; exit 0
fi
for t in "197 512" "37 2048"; do SSV_HIP_LIB=$OUT/libssv_attnstamp.so python tools/stamp_attn.py $t; done
