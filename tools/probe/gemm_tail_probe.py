#!/usr/bin/env python3
"""How expensive is a partial last round of workgroups?  Forward of a Linear layer 384 -> 1024 (8 column tiles) over M token rows, M around 98304 (= 768 x 128: whole
rounds of the 768 resident workgroups).   python tools/probe/gemm_tail_probe.py"""
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


n = 1024
w = (torch.randn(n, 1, 1, 384, device=dev) * 0.05).permute(0, 3, 1, 2)
b = torch.randn(n, device=dev)
for mt in (96, 192, 384, 576, 672, 768, 769, 776, 788, 800, 832, 864, 960, 1152, 1536, 1556):
    m = mt * 128
    x = torch.randn(m, 1, 1, 384, device=dev)
    t = timeit(lambda: ops.conv2d_fwd(x, w, 1, 0, bias=b))
    tiles = mt * (n // 128)
    print(f"M = {m:6d} ({mt:4d} row tiles): {t:.3f} ms  {2.0 * m * 384 * n / t / 1e9:6.1f} TFLOP/s   {tiles} workgroups = {tiles / 768:.2f} rounds of 768")
