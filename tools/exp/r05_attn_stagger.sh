#!/bin/bash
# r05: are the attention forward's workgroups in lock step (prologue + epilogue = one HBM burst per round, added to the MFMA loop)?  Persistent workgroups (plain per-item
# prologue), staggered at the start by class x SSV_ATTN_STAGGER sleeps of 8,128 cycles (class = (workgroup / 8) % 3); an item of the T = 197 launch is ~86 k cycles
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $2 SSV_HIP_LIB=$PWD/tools/probe/bin/libssv_attn_$3.so python3 tools/bench_attn.py 20 2>/dev/null | grep 'T  197\|T   37' | cut -c1-52 | tr '\n' '|')"; }
run "shipped            " "X=1" base
run "persistent, no stagger" "SSV_ATTN_STAGGER=0" stagger
run "persistent, stagger 2 " "SSV_ATTN_STAGGER=2" stagger
run "persistent, stagger 4 " "SSV_ATTN_STAGGER=4" stagger
run "persistent, stagger 7 " "SSV_ATTN_STAGGER=7" stagger
run "shipped            " "X=1" base
