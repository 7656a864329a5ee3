#!/usr/bin/env python3
"""Is the q/k/v GEMM (384 -> 1152 at 100864 tokens: 106 TFLOP/s against ~130 for its neighbours, tools/bench_vit_gemm.py) slow because of its output width?
Forward of a Linear layer 384 -> N over 100864 and 75776 token rows for N around 1152.   python tools/probe/gemm_width_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, rep=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep


for m in (100864, 98304, 75776):
    x = torch.randn(m, 1, 1, 384, device=dev)
    for n in (1024, 1152, 1280, 1536):
        w = (torch.randn(n, 1, 1, 384, device=dev) * 0.05).permute(0, 3, 1, 2)
        b = torch.randn(n, device=dev)
        t = timeit(lambda: ops.conv2d_fwd(x, w, 1, 0, bias=b))
        tiles = -(-m // 128) * -(-n // 128)
        print(f"tokens {m:6d}  384 -> {n:4d}: {t:.3f} ms  {2.0 * m * 384 * n / t / 1e9:6.1f} TFLOP/s   {tiles} workgroups = {tiles / 768:.2f} rounds of 768")
