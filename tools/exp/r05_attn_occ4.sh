#!/bin/bash
# r05: T = 197 attention forward at 4 waves per SIMD (key fragment read in pieces, 128 registers, 6 spilled) against the shipped 3 (141 registers) - tools/bench_attn.py, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "shipped"; python3 tools/bench_attn.py 20 2>/dev/null | grep "T  197\|T   37"
  echo "occ4";    SSV_HIP_LIB=$PWD/tools/probe/bin/libssv_attn_occ4.so python3 tools/bench_attn.py 20 2>/dev/null | grep "T  197\|T   37"
done
