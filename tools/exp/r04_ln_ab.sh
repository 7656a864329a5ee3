#!/bin/bash
# LayerNorm kernels: unit tests, then the probe on the shipped library and on the side library holding the previous kernels
set -e
python -m pytest tests/test_gpu_vit_ops.py -x -q -m gpu 2>&1 | tail -3
python tools/probe/ln_probe.py | tee gpurun_out/r04_probe_layernorm.txt
echo "--- previous kernels (three passes over x, one row per wave in flight)" | tee -a gpurun_out/r04_probe_layernorm.txt
SSV_HIP_LIB=tools/probe/bin/libssv_lnold.so python tools/probe/ln_probe.py | tee -a gpurun_out/r04_probe_layernorm.txt
