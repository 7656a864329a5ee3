#!/usr/bin/env python3
"""Where a wave of the one-pass attention backward (csrc/vit.hip attn_bwd_fused_k) spends a query tile's cycles: s_memtime stamps of a -DSSV_STAMP_ATTN diagnostic
build (vit.hip recompiled with the flag and linked with the shipped objects: tools/exp/r04_attn_stamps.sh).    SSV_HIP_LIB=<that library> python tools/stamp_attn.py [T] [B]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import _lib, ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 197
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
lib = ctypes.CDLL(_lib.LIB_PATH)
heads, hid = 6, 384
qkv = torch.randn(B * T, 3 * hid, device=dev)
q, k, v = qkv[:, :hid], qkv[:, hid:2 * hid], qkv[:, 2 * hid:]
dout = torch.randn(B * T, hid, device=dev)
o, lse = ops.attention_fwd(q, k, v, B, T, heads)
grads = torch.empty_like(qkv)
ops.attention_bwd(q, k, v, o, dout, lse, B, T, heads, out=grads)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.ssv_debug_attn_stamps(buf, 1)
ops.attention_bwd(q, k, v, o, dout, lse, B, T, heads, out=grads)
torch.cuda.synchronize()
lib.ssv_debug_attn_stamps(buf, 1)
n = buf[10]
names = ["issue next tile's loads", "fragment reads + S, dP (64 MFMAs)", "P, dS (16 exp2)", "dV, dK (64 MFMAs, 64 LDS operand reads)", "dS -> dS^T through LDS",
         "dQ partial (32 MFMAs)", "write the partial", "barrier 1", "reduce dQ: LDS reads, global stores", "barrier 2", None, "restage the next tile's rows"]
tot = sum(buf[i] for i in range(12) if i != 10)
print(f"T {T} B {B}: {n} stamped (wave, query tile) pairs (waves 0 and 5 of every workgroup)")
for i, name in enumerate(names):
    if name is None:
        continue
    print(f"   {name:42s} {buf[i] / n:9.0f} cycles / tile  {100.0 * buf[i] / tot:5.1f} %")
print(f"   total {tot / n:9.0f} cycles / tile; the wave's 160 MFMAs alone: 10240, its SIMD's two waves: 20480")
