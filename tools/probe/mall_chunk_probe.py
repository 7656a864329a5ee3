#!/usr/bin/env python3
"""Do the data gradient and the weight gradient of a 1x1 layer (256 <- 64 at 56x56, bs 512; both read the same 1.6 GB dY) get cheaper when they run
back to back on batch chunks small enough for dY to stay in the infinity cache?  Whole batch vs 16 x 32 / 8 x 64 / 4 x 128 images, interleaved per chunk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssv_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, rep=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
n, hw, c, k = 512, 56, 64, 256
x = torch.randn(n, hw, hw, c, device=dev)
dy = torch.randn(n, hw, hw, k, device=dev)
w = (torch.randn(k, 1, 1, c, device=dev) * 0.05).permute(0, 3, 1, 2)
dw = torch.zeros_like(w)
dx = torch.empty_like(x)
def whole():
    ops.conv2d_dgrad(dy, w, x.shape, 1, 0, out=None)
    ops.conv2d_wgrad(x, dy, w, dw, 1, 0, accumulate=True)
def chunked(cs):
    for a in range(0, n, cs):
        ops.conv2d_dgrad(dy[a:a + cs], w, x[a:a + cs].shape, 1, 0)
        ops.conv2d_wgrad(x[a:a + cs], dy[a:a + cs], w, dw, 1, 0, accumulate=True)
print(f"whole batch            : {timeit(whole):.3f} ms")
for cs in (128, 64, 32, 16):
    print(f"chunks of {cs:3d} images   : {timeit(lambda: chunked(cs)):.3f} ms")
