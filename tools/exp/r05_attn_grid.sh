#!/bin/bash
# r05: attention forward, grid (head, query block, image) + rotating last-tile wave against grid (query block, head, image) - tools/bench_attn.py, alternating
cd $GRAFT_REPO_ROOT
for v in base grid base grid base grid; do
  echo "$v: $(SSV_HIP_LIB=$PWD/tools/probe/bin/libssv_attn_$v.so python3 tools/bench_attn.py 20 2>/dev/null | grep 'T  197\|T   37' | cut -c1-52 | tr '\n' '|')"
done
