#!/bin/bash
# can HBM-bound kernels of one view run UNDER the other view's GEMMs if the GEMM workgroups leave registers free?  forward / data-gradient GEMM kernels padded to 57 KB of
# LDS (2 workgroups per CU instead of 3: 336 of 512 VGPRs per SIMD), tools/probe/build_variant.sh ldspad2 -DSSV_EXP_LDS_PAD=5120
for v in shipped ldspad2 shipped ldspad2; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  for ss in 0 1; do
    SSV_SINGLE_STREAM=$ss python bench.py --steps 20 --warmup 4 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v single_stream=$ss', 'images/s', d['value'], 'ms', d['ms_per_step'])"
  done
done
