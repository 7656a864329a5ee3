#!/usr/bin/env python3
"""Winograd F(2x2, 3x3) layers on BATCH CHUNKS: V (transformed input) and M (transformed-domain products) are pure intermediates, each 4x the tensor they come from -
written once and read once, ~150 GB/step of HBM traffic at bs 512.  On a chunk of the batch they are small enough to live in the 256 MB infinity cache between the
transform that writes them and the kernel that reads them.  Forward (fused input BatchNorm, statistics), gated data gradient and weight gradient (V kept) of the three
Winograd layer shapes, whole batch against chunks.        python tools/probe/wino_chunk_probe.py [batch = 512] [repeats = 5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


print(f"batch {B} per view, {REP} repeats; ms per layer and view (sum over the chunks)")
for name, H, Cc in (("p128.1.conv2 28x28x128", 28, 128), ("p256.1.conv2 14x14x256", 14, 256), ("p512.1.conv2 7x7x512", 7, 512)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, H, Cc, device=dev, generator=g)
    w = (torch.randn(Cc, Cc, 3, 3, device=dev, generator=g) * (2.0 / (9 * Cc)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, H, Cc, device=dev, generator=g)
    dw = torch.zeros_like(w)
    aff = (torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.1)
    mean, invstd = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5
    stats = H % 2 == 0
    for chunk in (B, B // 2, B // 4, B // 8, B // 16):
        spans = [(n0, min(n0 + chunk, B)) for n0 in range(0, B, chunk)]
        vs = {}

        def fwd():
            for n0, n1 in spans:
                _, _, v = ops.wino_conv2d_fwd(x[n0:n1], w, in_affine=aff, want_stats=stats, keep_v=True)
                vs[n0] = v

        def dgrad():
            for n0, n1 in spans:
                gate = ops.BnGateCtx(x[n0:n1], mean, invstd, scale=aff[0], shift=aff[1])
                ops.wino_conv2d_dgrad(dy[n0:n1], w, gate=gate)

        def wgrad():
            for n0, n1 in spans:
                ops.wino_conv2d_wgrad(vs[n0], dy[n0:n1], w, dw, accumulate=True)
        t_f = timeit(fwd)
        t_d = timeit(dgrad)
        t_w = timeit(wgrad)
        print(f"{name}  chunk {chunk:4d} x {len(spans):2d}: forward {t_f:.3f}  data gradient {t_d:.3f}  weight gradient {t_w:.3f}  | sum {t_f + t_d + t_w:.3f}", flush=True)
        vs.clear()
