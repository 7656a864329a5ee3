#!/usr/bin/env python3
"""Probe: Winograd F(2x2, 3x3) against the direct implicit-GEMM kernels on the stride-1 3x3 layers of ResNet-50 at bs 512 - time of each
stage (input transform, 16 batched GEMMs, output transform; data gradient; weight gradient from the kept V) and error against an fp64
convolution next to the direct kernel's.     python tools/probe_winograd.py [batch = 512] [repeats = 5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ssv_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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


class direct:
    """Inside: ops.* dispatch stays on the direct implicit-GEMM kernels (what the probe compares Winograd WITH)."""

    def __enter__(self):
        self.prev, ops.WINOGRAD = ops.WINOGRAD, False

    def __exit__(self, *exc):
        ops.WINOGRAD = self.prev
        return False


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print(f"batch {B}, {REP} repeats; times in ms, TFLOP/s of the direct convolution's algorithmic work")
for name, H, Cc in (("p64.0.conv2", 56, 64), ("p128.1.conv2", 28, 128), ("p256.1.conv2", 14, 256), ("p512.1.conv2", 7, 512)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, H, Cc, device=dev, generator=g)
    w = (torch.randn(Cc, Cc, 3, 3, device=dev, generator=g) * (2.0 / (9 * Cc)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, H, Cc, device=dev, generator=g)
    aff = (torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.1)
    flop = 2.0 * B * H * H * Cc * Cc * 9
    tf = lambda ms: flop / (ms * 1e-3) / 1e12
    # ---- forward (fused input BatchNorm + statistics epilogue, as the step launches it)
    stats = int(_lib.load().ssv_wino_stats_rows_per_group(B, H, H)) > 0
    with direct():
        t_dir = timeit(lambda: ops.conv2d_fwd_fused(x, w, 1, 1, in_affine=aff, want_stats=True))
    t_win = timeit(lambda: ops.wino_conv2d_fwd(x, w, in_affine=aff, want_stats=stats, keep_v=True))
    lib = _lib.load()
    n, h, w_, c = x.shape
    T = int(lib.ssv_wino_tiles(n, h, w_))
    u = ops._wino_filter(w, tuple(w.shape))
    v = torch.empty((16, T, Cc), device=dev)
    m = torch.empty((16, T, Cc), device=dev)
    y = torch.empty((n, h, w_, Cc), device=dev)
    part = torch.empty((2, int(lib.ssv_wino_groups(n, h, w_)), Cc), device=dev)
    t_in = timeit(lambda: _lib.call("ssv_wino_input_transform", n, h, w_, Cc, _lib.ptr(x), _lib.ptr(aff[0]), _lib.ptr(aff[1]), _lib.ptr(v), _lib.stream()))
    t_gm = timeit(lambda: _lib.call("ssv_gemm_batched", 16, T, Cc, Cc, _lib.ptr(v), _lib.ptr(u), _lib.ptr(m), _lib.stream()))
    t_out = timeit(lambda: _lib.call("ssv_wino_output_transform", n, h, w_, Cc, _lib.ptr(m), _lib.ptr(y), _lib.ptr(part[0]) if stats else None, _lib.ptr(part[1]) if stats else None, None, _lib.stream()))
    print(f"{name:14s} fwd   direct {t_dir:6.3f} ({tf(t_dir):5.1f} TF)  winograd {t_win:6.3f} ({tf(t_win):5.1f} TF) = in {t_in:.3f} + gemm {t_gm:.3f} ({flop / 2.25 * (T * 4 / (B * H * H)) / (t_gm * 1e-3) / 1e12:5.1f} TF executed) + out {t_out:.3f}   x{t_dir / t_win:.2f}")
    # ---- data gradient (recomputed-gate epilogue, as behind conv2 of a bottleneck)
    gx = torch.randn(B, H, H, Cc, device=dev, generator=g)
    mean, invstd = torch.randn(Cc, device=dev, generator=g) * 0.1, torch.rand(Cc, device=dev, generator=g) + 0.5
    gate = ops.BnGateCtx(gx, mean, invstd, scale=aff[0], shift=aff[1])
    with direct():
        t_dd = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1, gate=gate))
    t_dw = timeit(lambda: ops.wino_conv2d_dgrad(dy, w, gate=gate))
    print(f"{'':14s} dgrad direct {t_dd:6.3f} ({tf(t_dd):5.1f} TF)  winograd {t_dw:6.3f} ({tf(t_dw):5.1f} TF)   x{t_dd / t_dw:.2f}")
    # ---- weight gradient (V kept from the forward)
    dw = torch.zeros_like(w)
    _, _, vkeep = ops.wino_conv2d_fwd(x, w, in_affine=aff, want_stats=False, keep_v=True)
    t_wd = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, 1, 1, accumulate=True, in_affine=aff))
    t_ww = timeit(lambda: ops.wino_conv2d_wgrad(vkeep, dy, w, dw, accumulate=True))
    print(f"{'':14s} wgrad direct {t_wd:6.3f} ({tf(t_wd):5.1f} TF)  winograd {t_ww:6.3f} ({tf(t_ww):5.1f} TF)   x{t_wd / t_ww:.2f}")
    # ---- error against fp64 on the first samples
    ns = min(B, 8)
    xs = torch.relu(x[:ns].double() * aff[0].double() + aff[1].double()).cpu()
    ws_ = w.double().cpu()
    ref = F.conv2d(xs.permute(0, 3, 1, 2), ws_, padding=1).permute(0, 2, 3, 1)
    with direct():
        yd, _ = ops.conv2d_fwd_fused(x[:ns].contiguous(), w, 1, 1, in_affine=aff, want_stats=True)
    yw, pw, _ = ops.wino_conv2d_fwd(x[:ns].contiguous(), w, in_affine=aff, want_stats=stats)
    line = f"{'':14s} error vs fp64 (rel l2): fwd direct {rel(yd.cpu(), ref):.2e} winograd {rel(yw.cpu(), ref):.2e}"
    dys = dy[:ns].double().cpu()
    refdx = F.conv_transpose2d(dys.permute(0, 3, 1, 2), ws_, padding=1).permute(0, 2, 3, 1)
    with direct():
        dxd = ops.conv2d_dgrad(dy[:ns].contiguous(), w, (ns, H, H, Cc), 1, 1)
    dxw = ops.wino_conv2d_dgrad(dy[:ns].contiguous(), w)
    line += f" | dgrad direct {rel(dxd.cpu(), refdx):.2e} winograd {rel(dxw.cpu(), refdx):.2e}"
    xr = xs.permute(0, 3, 1, 2).clone().requires_grad_(False)
    wr = ws_.clone().requires_grad_()
    F.conv2d(xr, wr, padding=1).backward(dys.permute(0, 3, 1, 2))
    dwd, dww = torch.zeros_like(w), torch.zeros_like(w)
    ops.conv2d_wgrad(x[:ns].contiguous(), dy[:ns].contiguous(), w, dwd, 1, 1, accumulate=False, in_affine=aff)
    _, _, vk = ops.wino_conv2d_fwd(x[:ns].contiguous(), w, in_affine=aff, keep_v=True)
    ops.wino_conv2d_wgrad(vk, dy[:ns].contiguous(), w, dww, accumulate=False)
    line += f" | wgrad direct {rel(dwd.cpu(), wr.grad):.2e} winograd {rel(dww.cpu(), wr.grad):.2e}"
    print(line, flush=True)
    if stats:      # the statistics partials of the output transform against the direct kernel's, merged
        with direct():
            _, pd = ops.conv2d_fwd_fused(x[:ns].contiguous(), w, 1, 1, in_affine=aff, want_stats=True)
        m_rows = ns * H * H

        def mrg(pm, p2, rpg):          # Chan merge of the per-group (mean, M2) partials; the last group may be ragged
            cnt = torch.full((pm.shape[0], 1), float(rpg), dtype=torch.float64, device=pm.device)
            cnt[-1] = m_rows - rpg * (pm.shape[0] - 1)
            mean = (pm.double() * cnt).sum(0) / m_rows
            return mean, p2.double().sum(0) + (cnt * (pm.double() - mean) ** 2).sum(0)
        md, vd = mrg(pd[0], pd[1], 64)
        mw, vw = mrg(pw[0], pw[1], pw[2])
        print(f"{'':14s} statistics partials: mean diff {float((md - mw).abs().max()):.2e}, M2 rel diff {float(((vd - vw).abs() / vd).max()):.2e}")
    del x, w, dy, v, m, y, vkeep
