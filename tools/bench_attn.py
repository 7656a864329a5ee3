#!/usr/bin/env python3
"""Attention kernels alone on the DINO ViT-S/16 multi-crop shapes (bs 128): 37-token local crops (B = 2048 per student pass), 197-token
global crops (B = 512), 6 heads of 64 - forward and backward time, TFLOP/s of algorithmic work (4 T^2 d per head forward, 8 T^2 d backward).
    python tools/bench_attn.py [repeats = 10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


for T, B in ((37, 2048), (197, 512), (48, 2048), (64, 1024)):
    heads, hid = 6, 384
    qkv = torch.randn(B * T, 3 * hid, device=dev)
    q, k, v = qkv[:, :hid], qkv[:, hid:2 * hid], qkv[:, 2 * hid:]
    dout = torch.randn(B * T, hid, device=dev)
    o, lse = ops.attention_fwd(q, k, v, B, T, heads)
    grads = torch.empty_like(qkv)
    t_f = timeit(lambda: ops.attention_fwd(q, k, v, B, T, heads))
    t_b = timeit(lambda: ops.attention_bwd(q, k, v, o, dout, lse, B, T, heads, out=grads))
    flop = 4.0 * B * heads * T * T * 64
    print(f"T {T:4d} B {B:5d}: forward {t_f:7.3f} ms ({flop / t_f / 1e9:6.1f} TFLOP/s)   backward {t_b:7.3f} ms ({2 * flop / t_b / 1e9:6.1f} TFLOP/s)", flush=True)
