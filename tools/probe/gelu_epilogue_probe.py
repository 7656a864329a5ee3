#!/usr/bin/env python3
"""What do the GELU epilogues of the ViT FFN cost?  fc1 (384 -> 1536) forward: plain bias epilogue vs + gelu (h and gelu(h) written / gelu(h) only); fc2's data gradient
(1536 <- 384): plain vs x gelu'(h).  Both DINO token counts.      python tools/probe/gelu_epilogue_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, rep=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep


x0 = torch.randn(100864, 384, device=dev)
w0 = torch.randn(1536, 384, device=dev) * 0.05
for _ in range(50):                                   # warm the clocks before the first measurement
    ops.linear_gelu_fwd(x0, w0, None)
for m in (100864, 75776):
    x = torch.randn(m, 384, device=dev)
    w1 = torch.randn(1536, 384, device=dev) * 0.05
    b1 = torch.randn(1536, device=dev)
    w2 = torch.randn(384, 1536, device=dev) * 0.05
    dy = torch.randn(m, 384, device=dev)
    h = torch.randn(m, 1536, device=dev)
    x4, w14 = x.view(m, 1, 1, 384), w1.view(1536, 384, 1, 1).contiguous(memory_format=torch.channels_last)
    t_plain = timeit(lambda: ops.conv2d_fwd(x4, w14, 1, 0, bias=b1))
    t_both = timeit(lambda: ops.linear_gelu_fwd(x, w1, b1, keep_h=True))
    t_act = timeit(lambda: ops.linear_gelu_fwd(x, w1, b1, keep_h=False))
    dy4, w24 = dy.view(m, 1, 1, 384), w2.view(384, 1536, 1, 1).contiguous(memory_format=torch.channels_last)
    t_dplain = timeit(lambda: ops.conv2d_dgrad(dy4, w24, (m, 1, 1, 1536), 1, 0))
    t_dgelu = timeit(lambda: ops.linear_dgrad_gelu(dy, w2, h))
    gf = 2.0 * m * 384 * 1536 / 1e9
    print(f"tokens {m}: fc1 forward plain {t_plain:.3f} ms ({gf / t_plain:.1f} TF) | + h and gelu(h) {t_both:.3f} | gelu(h) only {t_act:.3f}   ||   "
          f"fc2 data gradient plain {t_dplain:.3f} ms ({gf / t_dplain:.1f} TF) | x gelu'(h) {t_dgelu:.3f}")
