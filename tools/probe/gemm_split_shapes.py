#!/usr/bin/env python3
"""The transformed-domain products of the Winograd layers at batch 512 (36 positions; tiles x channels of the 28^2 / 14^2 / 7^2 stages) on the fp32-MFMA batched GEMM
and on the bf16-split one (6 / 9 terms): ms per launch, TFLOP/s of 2 * rows * C * K * 36, against the bytes each moves (a + y once, w negligible).
    python tools/probe/gemm_split_shapes.py [repeats = 20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssv_amd import _lib

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


for nb, t, c, k in ((36, 25088, 128, 128), (36, 8192, 256, 256), (36, 2048, 512, 512), (36, 100352, 64, 64), (16, 8192, 1024, 1024)):
    a = torch.randn(nb, t, c, device=dev)
    w = torch.randn(nb, k, c, device=dev) * 0.05
    y = torch.empty(nb, t, k, device=dev)
    flop = 2.0 * nb * t * c * k
    gb = 4.0 * nb * (t * c + t * k + k * c) / 1e9
    line = f"{nb} x [{t} x {c}] . [{k} x {c}]^T  ({flop / 1e9:.1f} GFLOP, {gb:.2f} GB = {gb / 6.29:.3f} ms at 6.29 TB/s):"
    ms = timeit(lambda: _lib.call("ssv_gemm_batched", nb, t, c, k, _lib.ptr(a), _lib.ptr(w), _lib.ptr(y), _lib.stream()))
    line += f"  fp32 MFMA {ms:.3f} ms ({flop / ms / 1e9:.0f} TFLOP/s)"
    for terms in (6, 9):
        ms = timeit(lambda: _lib.call("ssv_gemm_batched_split", nb, t, c, k, _lib.ptr(a), _lib.ptr(w), _lib.ptr(y), terms, _lib.stream()))
        line += f" | split {terms}: {ms:.3f} ms ({flop / ms / 1e9:.0f})"
    print(line, flush=True)
