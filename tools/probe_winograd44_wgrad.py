#!/usr/bin/env python3
"""Probe (round 5): the WEIGHT gradient of the Winograd layers of ResNet-50 at bs 512 through F(4x4, 3x3) against F(2x2, 3x3) and the direct kernel - time of the
whole product (dY transform + batched GEMMs + filter back-transform), of the forward's input transform with and without the second (F(2x2)) operand, and the
relative l2 error against an fp64 weight gradient of the FULL batch (nine fp64 GEMMs on the GPU: test infrastructure, torch).
    python tools/probe_winograd44_wgrad.py [batch = 512] [repeats = 5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
CHUNK0, FLUSH0 = ops.WINOGRAD44_WGRAD_CHUNK, ops.WINOGRAD44_WGRAD_FLUSH


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


def wgrad64(a, dy):
    """fp64 weight gradient [K][C][3][3] of a 3x3 / padding 1 convolution from NHWC operands, as nine GEMMs."""
    n, h, w_, c = a.shape
    k = dy.shape[3]
    ap = torch.zeros((n, h + 2, w_ + 2, c), dtype=torch.float64, device=a.device)
    ap[:, 1:-1, 1:-1] = a.double()
    d2 = dy.double().reshape(-1, k)
    out = torch.empty((k, c, 3, 3), dtype=torch.float64, device=a.device)
    for r in range(3):
        for s in range(3):
            out[:, :, r, s] = d2.t() @ ap[:, r:r + h, s:s + w_].reshape(-1, c)
    return out


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print(f"batch {B}, {REP} repeats; ms per layer and view; error = relative l2 of dW against an fp64 weight gradient of the full batch")
for name, H, Cc in (("p128.1.conv2 28x28x128", 28, 128), ("p256.1.conv2 14x14x256", 14, 256), ("p512.1.conv2 7x7x512", 7, 512)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, H, Cc, device=dev, generator=g)
    w = (torch.randn(Cc, Cc, 3, 3, device=dev, generator=g) * (2.0 / (9 * Cc)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, H, Cc, device=dev, generator=g)
    aff = (torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.1)
    a = torch.relu(x * aff[0] + aff[1])                        # what the layer really sees: a BatchNorm + ReLU output
    ref = wgrad64(a, dy)
    lib = ops._lib.load()
    dw = torch.zeros_like(w)
    # direct kernel
    t_dir = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, 1, 1, accumulate=False, in_affine=aff))
    e_dir = rel(dw.permute(0, 1, 2, 3), ref)
    # F(2x2): its operand from F(2x2)'s own input transform
    v2 = torch.empty((16, int(lib.ssv_wino_tiles(B, H, H)), Cc), device=dev)
    ops.call("ssv_wino_input_transform", B, H, H, Cc, ops.ptr(x), ops.ptr(aff[0]), ops.ptr(aff[1]), ops.ptr(v2), ops.stream())
    t_22 = timeit(lambda: ops.wino_conv2d_wgrad(v2, dy, w, dw, accumulate=False))
    e_22 = rel(dw, ref)
    # F(4x4)
    t44 = int(lib.ssv_wino44_tiles(B, H, H))
    v4 = torch.empty((36, t44, Cc), device=dev)
    both = lambda: ops.call("ssv_wino44_input_transform", B, H, H, Cc, ops.ptr(x), ops.ptr(aff[0]), ops.ptr(aff[1]), ops.ptr(v4), ops.ptr(v2), ops.stream())
    only = lambda: ops.call("ssv_wino44_input_transform", B, H, H, Cc, ops.ptr(x), ops.ptr(aff[0]), ops.ptr(aff[1]), ops.ptr(v4), None, ops.stream())
    t_in_both, t_in_only = timeit(both), timeit(only)
    sweep = []
    for chunk in (0, 2048, 1024, 512, 256, 128):
        ops.WINOGRAD44_WGRAD_CHUNK = chunk
        sweep.append((chunk, timeit(lambda: ops.wino44_conv2d_wgrad(v4, dy, w, dw, accumulate=False)), rel(dw, ref)))
    ops.WINOGRAD44_WGRAD_FLUSH = 128
    for chunk in (0, 2048, 1024, 512):
        ops.WINOGRAD44_WGRAD_CHUNK = chunk
        sweep.append((f"flush128+{chunk}", timeit(lambda: ops.wino44_conv2d_wgrad(v4, dy, w, dw, accumulate=False)), rel(dw, ref)))
    ops.WINOGRAD44_WGRAD_CHUNK, ops.WINOGRAD44_WGRAD_FLUSH = CHUNK0, FLUSH0
    t_44 = timeit(lambda: ops.wino44_conv2d_wgrad(v4, dy, w, dw, accumulate=False))
    e_44 = rel(dw, ref)
    dm = torch.empty((36, t44, Cc), device=dev)
    t_dy44 = timeit(lambda: ops.call("ssv_wino44_dy_transform", B, H, H, Cc, ops.ptr(dy), ops.ptr(dm), ops.stream()))
    dm2 = torch.empty((16, v2.shape[1], Cc), device=dev)
    t_dy22 = timeit(lambda: ops.call("ssv_wino_dy_transform", B, H, H, Cc, ops.ptr(dy), ops.ptr(dm2), ops.stream()))
    print(f"{name}  direct {t_dir:.3f} ms err {e_dir:.2e} | F(2x2) {t_22:.3f} ms (dY transform {t_dy22:.3f}) err {e_22:.2e} | F(4x4) {t_44:.3f} ms (dY transform {t_dy44:.3f}) err {e_44:.2e}"
          f" | forward input transform with / without the F(2x2) operand {t_in_both:.3f} / {t_in_only:.3f} ms", flush=True)
    print("    F(4x4) by accumulation chunk (tiles per fp32 chain; 0 = plain split): " + "  ".join(f"{c}: {t:.3f} ms {e:.2e}" for c, t, e in sweep), flush=True)
