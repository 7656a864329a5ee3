#!/bin/bash
# r05: one-pass attention backward, the rows of the tile after the next asked for BEFORE this tile's dQ stores (new) against at the top of the next trip (old) - alternating
cd $GRAFT_REPO_ROOT
for v in old new old new old new; do
  echo "$v: $(SSV_HIP_LIB=$PWD/tools/probe/bin/libssv_attn_bwd_$v.so python3 tools/bench_attn.py 20 2>/dev/null | grep 'T  197\|T   64' | cut -c53-100 | tr '\n' '|')"
done
