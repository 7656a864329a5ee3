#!/bin/bash
# r05: the persistent attention forward (next item's K / V tile and Q rows fetched under the current item's last key tile) in four builds - 3 or 2 workgroups per CU, the
# next Q rows fetched behind the last S product or behind the last P V product - against the shipped kernel (one item per workgroup): tools/bench_attn.py
cd $GRAFT_REPO_ROOT
for v in shipped p3 p2 late3 late2 shipped; do
  if [ $v = shipped ]; then L=$PWD/tools/probe/bin/libssv_attn_base.so; else L=$PWD/tools/probe/bin/libssv_attn_$v.so; fi
  echo "$v: $(SSV_HIP_LIB=$L python3 tools/bench_attn.py 20 2>/dev/null | grep 'T  197\|T   37' | cut -c1-52 | tr '\n' '|')"
done
