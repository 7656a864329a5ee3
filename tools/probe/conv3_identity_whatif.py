#!/usr/bin/env python3
"""What-if for the conv3-backward identity: how much of a stage-1 / stage-2 unit's backward would it save at best?

Today (per unit and view): the gate epilogue that produces g reads c3 for sum g * xhat; conv3's weight gradient and data gradient form
dx3 = A g + B (c3 - mean) + D on load from (g, c3).  The identity replaces every read of c3 [M x 4p] by reads of a2 [M x p]:
    G = g^T a2 (weight-gradient GEMM, plain dY),  Gram = a2^T a2 (one more, p x p),  dX2 = [g | a2] . [diag(A) W3 ; W3^T diag(B) W3]  (K = 5p instead of 4p).
Every piece of the new form exists as a kernel shape today, so it can be timed WITHOUT building the identity:
    G      = conv2d_wgrad(x = c2 with the fused BatchNorm input, dy = g plain)
    Gram   = conv2d_wgrad(x = a2, dy = a2) on a materialised a2 (+ the pass that materialises it), the cheapest form with today's loaders
    dX2    = the gated 1x1 data gradient with a contraction of 5p contiguous channels (same bytes and FLOPs as the two-source loader [g | a2])
    python tools/probe/conv3_identity_whatif.py [batch = 512] [repeats = 10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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


print(f"batch {B} per view, {REP} repeats, ms per launch")
total = 0.0
for name, H, p, units in (("stage 1 (56x56, 64 -> 256)", 56, 64, 3), ("stage 2 (28x28, 128 -> 512)", 28, 128, 4)):
    K = 4 * p
    M = B * H * H
    c2 = torch.randn(B, H, H, p, device=dev)
    g = torch.randn(B, H, H, K, device=dev)
    c3 = torch.randn(B, H, H, K, device=dev)
    w3 = (torch.randn(K, p, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros_like(w3)
    aff = (torch.rand(p, device=dev) + 0.5, torch.randn(p, device=dev) * 0.1)
    coef = torch.randn(4, K, device=dev).contiguous()
    lazy = ops.LazyGrad(g, c3, coef)
    mean, invstd = torch.randn(p, device=dev) * 0.1, torch.rand(p, device=dev) + 0.5
    gate = ops.BnGateCtx(c2, mean, invstd, scale=aff[0], shift=aff[1])
    # ---- today
    t_w_now = timeit(lambda: ops.conv2d_wgrad(c2, lazy, w3, dw, 1, 0, accumulate=True, in_affine=aff))
    t_d_now = timeit(lambda: ops.conv2d_dgrad(lazy, w3, c2.shape, 1, 0, gate=gate))
    # ---- the identity's pieces
    t_G = timeit(lambda: ops.conv2d_wgrad(c2, g, w3, dw, 1, 0, accumulate=True, in_affine=aff))
    a2 = torch.randn(B, H, H, p, device=dev)
    wpp = (torch.randn(p, p, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dpp = torch.zeros_like(wpp)
    t_gram = timeit(lambda: ops.conv2d_wgrad(a2, a2, wpp, dpp, 1, 0, accumulate=True))
    t_gram_fused = timeit(lambda: ops.conv2d_wgrad(c2, a2, wpp, dpp, 1, 0, accumulate=True, in_affine=aff))   # x operand formed on load, dY still materialised
    t_mat = timeit(lambda: a2.copy_(c2))                                     # stand-in for the BatchNorm-apply pass that writes a2 (1 read + 1 write of [M x p])
    g5 = torch.randn(B, H, H, 5 * p, device=dev)
    w5 = (torch.randn(5 * p, p, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    t_d_new = timeit(lambda: ops.conv2d_dgrad(g5, w5, c2.shape, 1, 0, gate=gate))
    gate_read_ms = 4.0 * M * K / 5.0e12 * 1e3                               # the gate epilogue's read of c3 at 5 TB/s (what those HBM-bound launches reach)
    new = t_G + t_gram + t_mat + t_d_new
    now = t_w_now + t_d_now + gate_read_ms
    print(f"{name}: today  wgrad(dy formed on load) {t_w_now:.3f} + dgrad(dy formed on load, gated) {t_d_now:.3f} + gate's c3 read ~{gate_read_ms:.3f} = {now:.3f}")
    print(f"{' ' * len(name)}  identity  G {t_G:.3f} + Gram {t_gram:.3f} (x formed on load: {t_gram_fused:.3f}) + materialise a2 {t_mat:.3f} + dgrad over [g | a2] {t_d_new:.3f} = {new:.3f}"
          f"   -> saves {now - new:+.3f} ms per unit and view, x {units} units x 2 views = {(now - new) * units * 2:+.2f} ms/step (upper bound: the tiny combine kernels are not counted)")
    total += (now - new) * units * 2
    del c2, g, c3, g5, a2
print(f"upper bound of the identity over both stages: {total:+.2f} ms/step")
