#!/bin/bash
# r05: what the T = 197 attention forward (attn_fwd_k<4>, 0.38 ms, matrix pipe 0.64 busy) spends beside its MFMAs: diagnostic builds with one piece removed each
# (-DSSV_ATTN_WHATIF: 1 no softmax arithmetic, 2 PV operands from registers, 4 S operand from registers, 8 no barriers / restaging, 15 all four, 16 ONE key tile only = prologue + one tile + epilogue, 31 = 16 + 15) - tools/bench_attn.py
cd $GRAFT_REPO_ROOT
for w in 0 1 2 4 8 15 16 31 0; do
  echo "whatif $w: $(SSV_HIP_LIB=$PWD/tools/probe/bin/libssv_attn_w$w.so python3 tools/bench_attn.py 20 2>/dev/null | grep 'T  197' | cut -c1-60)"
done
