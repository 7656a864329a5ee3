#!/usr/bin/env python3
"""Probe: Winograd F(4x4, 3x3) (csrc/winograd44.hip) against F(2x2, 3x3) and the direct implicit-GEMM kernels on the Winograd layers of ResNet-50 at bs 512 - forward
(fused input BatchNorm, statistics epilogue, V2 kept) and gated data gradient: time of each, error against an fp64 convolution next to the other two, and the check
that the F(2x2) operand the F(4x4) input transform leaves is bitwise the one F(2x2)'s own transform writes.      python tools/probe_winograd44.py [batch = 512] [repeats = 5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

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


def rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm())


class mode:
    def __init__(self, wino, w44):
        self.want = (wino, w44)

    def __enter__(self):
        self.prev = (ops.WINOGRAD, ops.WINOGRAD44)
        ops.WINOGRAD, ops.WINOGRAD44 = self.want

    def __exit__(self, *exc):
        ops.WINOGRAD, ops.WINOGRAD44 = self.prev
        return False


print(f"batch {B}, {REP} repeats; ms per layer and view; error = relative l2 against an fp64 convolution of a {min(B, 16)}-image slice")
for name, H, Cc in (("p128.1.conv2 28x28x128", 28, 128), ("p256.1.conv2 14x14x256", 14, 256), ("p512.1.conv2 7x7x512", 7, 512)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, H, Cc, device=dev, generator=g)
    w = (torch.randn(Cc, Cc, 3, 3, device=dev, generator=g) * (2.0 / (9 * Cc)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, H, Cc, device=dev, generator=g)
    aff = (torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.1)
    mean, invstd = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5
    gate = ops.BnGateCtx(x, mean, invstd, scale=aff[0], shift=aff[1])
    ns = min(B, 16)
    a64 = torch.relu(x[:ns].double().cpu() * aff[0].double().cpu() + aff[1].double().cpu())
    ref_f = F.conv2d(a64.permute(0, 3, 1, 2), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
    ref_d = F.conv_transpose2d(dy[:ns].double().cpu().permute(0, 3, 1, 2), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
    out = {}
    for tag, (wi, w4) in (("direct", (False, False)), ("F(2x2)", (True, False)), ("F(4x4)", (True, True))):
        with mode(wi, w4):
            t_f = timeit(lambda: ops.conv2d_fwd_fused(x, w, 1, 1, in_affine=aff, want_stats=True, keep_v=True))
            t_d = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1, gate=gate))
            yf = ops.conv2d_fwd_fused(x[:ns], w, 1, 1, in_affine=aff, want_stats=False)[0]
            dx = ops.conv2d_dgrad(dy[:ns], w, x[:ns].shape, 1, 1)
            out[tag] = (t_f, t_d, rel(yf, ref_f), rel(dx, ref_d))
    for tag, (t_f, t_d, e_f, e_d) in out.items():
        print(f"{name}  {tag:7s} forward {t_f:.3f} ms  data gradient {t_d:.3f} ms   error fwd {e_f:.2e} ({e_f / out['direct'][2]:.2f}x direct)  dgrad {e_d:.2e} ({e_d / out['direct'][3]:.2f}x direct)")
    with mode(True, False):
        v2a = ops.wino_conv2d_fwd(x, w, in_affine=aff, want_stats=False, keep_v=True)[2]
    with mode(True, True):
        y4, part, v2b = ops.wino_conv2d_fwd(x, w, in_affine=aff, want_stats=True, keep_v=True)
    print(f"{' ' * len(name)}  F(2x2) operand left by the F(4x4) input transform bitwise equal to F(2x2)'s own: {torch.equal(v2a, v2b)}; statistics groups {tuple(part[0].shape)}, {part[2]} rows each", flush=True)
