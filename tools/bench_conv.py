#!/usr/bin/env python3
"""Per-layer table of the conv implicit-GEMM kernels on the ResNet-50 @224 shapes (all 53 convolutions, identical shapes merged):
for forward, data gradient and weight gradient - the variant the training step really launches (statistics epilogue, fused input
BatchNorm, gated epilogue; the Winograd form and the stem kernel are read off the dispatcher / the library, not assumed), its time, TFLOP/s of algorithmic work, and how its grid quantises onto the chip
(workgroups / (3 resident per CU x 256 CUs) = "waves of 768").

    python tools/bench_conv.py [batch per view = 512] [repeats = 5] [out.csv]
SSV_BENCH_LAYERS=p64.,p128.0 restricts the table to layers whose name contains one of the comma-separated substrings.
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import _lib, ops  # noqa: E402
from ssv_amd import nn as hnn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
OUT = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")


def resnet50_shapes():
    shapes = [("stem", 224, 3, 64, 7, 2, 3, "plain")]            # name, H, C, K, R, stride, pad, input kind
    cin, h = 64, 56
    for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for b in range(blocks):
            s = stride if b == 0 else 1
            shapes.append((f"p{planes}.{b}.conv1", h, cin, planes, 1, 1, 0, "plain"))
            shapes.append((f"p{planes}.{b}.conv2", h, planes, planes, 3, s, 1, "lazy"))
            h2 = (h + 2 - 3) // s + 1
            shapes.append((f"p{planes}.{b}.conv3", h2, planes, planes * 4, 1, 1, 0, "lazy"))
            if b == 0:
                shapes.append((f"p{planes}.{b}.downsample", h, cin, planes * 4, 1, s, 0, "plain"))
            cin, h = planes * 4, h2
    uniq = {}
    for n, H, C, K, R, s, p, kind in shapes:
        key = (H, C, K, R, s, p, kind)
        if key in uniq:
            uniq[key][1] += 1
        else:
            uniq[key] = [n, 1]
    return uniq


def xin_b(b, h, c):
    return 4.0 * b * h * h * c


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


def wino_form(fn):
    """Which Winograd form(s) a call dispatches to, from the dispatcher's own log (ops.DISPATCH): '' | 'F(4x4)' | 'F(2x2)' | both."""
    prev, ops.DISPATCH = ops.DISPATCH, {}
    try:
        fn()
        keys = set(ops.DISPATCH)
    finally:
        ops.DISPATCH = prev
    forms = [f for f, tag in (("F(4x4)", "wino44_"), ("F(2x2)", "wino22_")) if any(k.startswith(tag) for k in keys)]
    return "+".join(forms)


def grid_fwd(m, k):
    return -(-m // 128) * -(-k // 128) if k >= 128 else -(-m // 256) * -(-k // 64)


rows, tot = [], {}
hdr = ["layer", "count", "HxW", "C", "K", "RxS", "stride", "GFLOP", "fwd_variant", "fwd_ms", "fwd_TF", "fwd_wgs", "fwd_waves768",
       "dgrad_variant", "dgrad_ms", "dgrad_TF", "dgrad_wgs", "wgrad_variant", "wgrad_ms", "wgrad_TF",
       "fwd_GB", "fwd_TBs", "fwd_frac", "dgrad_GB", "dgrad_TBs", "dgrad_frac", "wgrad_GB", "wgrad_TBs", "wgrad_frac"]
# a layer's own bound = max(FLOP / matrix-pipe peak of the arithmetic the launch runs on, bytes / HBM achievable) (MI355X_MICROARCH.md): fp32 MFMA dense 157.3 TFLOP/s;
# SSV_ARITH_BF16X3 (round 6): dense bf16 2,500 TFLOP/s / 6 piece products per fp32 product = 416.7 TFLOP/s of fp32 products
FP32_PEAK, BF16X3_PEAK, HBM_PEAK = 157.3e12, 2500e12 / 6, 6.29e12


def pipe_peak(x_shape, w_shape, stride, pad, product):
    """The roof of the pipe this product of this layer runs on: asks the library (ssv_conv_arithmetic) what the descriptor the step builds will run."""
    if ops.ARITHMETIC != "bf16x3" or w_shape[1] == 3:
        return FP32_PEAK
    import ctypes
    d = ops.conv_desc(tuple(x_shape), tuple(w_shape), stride, pad)
    d.arithmetic = _lib.ARITH_BF16X3
    d.w_planes = 16                                            # any non-null address: the query does not dereference it
    return BF16X3_PEAK if _lib.load().ssv_conv_arithmetic(ctypes.byref(d), product) == _lib.ARITH_BF16X3 else FP32_PEAK


def roof(flop, nbytes, ms, peak):
    """(GB moved when every operand stream of the variant is moved exactly once, TB/s achieved, fraction of the layer's own roofline bound)."""
    if ms != ms:
        return "", "", ""
    bound = max(flop / peak, nbytes / HBM_PEAK)
    return round(nbytes / 1e9, 2), round(nbytes / (ms * 1e-3) / 1e12, 2), round(bound / (ms * 1e-3), 3)

print(f"ResNet-50 @224, batch {B} per view, {REP} repeats")
print(" ".join(f"{h:>13s}" for h in hdr))
ONLY = [t for t in os.environ.get("SSV_BENCH_LAYERS", "").split(",") if t]
for (H, C, K, R, s, p, kind), (name, cnt) in resnet50_shapes().items():
    if ONLY and not any(t in name for t in ONLY):
        continue
    if C == 3:                        # the image stem: row-taps form on the unpadded image (nn.stem_conv)
        x = torch.randn(B, H, H, 3, device=dev)
        w = (torch.randn(K, 3, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        Cx = 3
    else:
        x = torch.randn(B, H, H, C, device=dev)
        w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        Cx = C
    y = ops.stem_conv_fwd(x, ops.stem_weight_rows(w), tuple(w.shape), s, p)[0] if C == 3 else ops.conv2d_fwd(x, w, s, p)
    Ho = y.shape[1]
    m = B * Ho * Ho
    flop = 2.0 * y.numel() * C * R * R                        # algorithmic: the 3 real channels for the stem
    dy = torch.randn_like(y)
    dw = torch.zeros_like(w)
    aff = (torch.rand(Cx, device=dev) + 0.5, torch.randn(Cx, device=dev) * 0.1)
    stats_ok = Cx % 32 == 0 and K % 4 == 0
    # what the training step hands these layers (nn.py): dY formed on load behind 1x1 / stride-1 convolutions on maps of >= 28x28, and the
    # closing activation of the previous unit formed and written by conv1 on such maps (every conv1 but the first one, whose input is the pool's)
    # the step hands a k x k convolution with fewer than 128 output channels a MATERIALISED activation (nn.HipConv2d.can_fuse_input: re-transforming each element
    # k*k times costs more than the apply pass saves there): layer1's 3x3 runs the plain statistics variant, its gate reads the byte mask
    # ... unless that narrow layer runs Winograd (round 5: 64 channels on F(4x4)): its input transform forms BatchNorm + ReLU on load (nn.HipConv2d.can_fuse_input)
    lazy_in = kind == "lazy" and ops.can_fuse_conv_input(Cx, K) and (not (R > 1 and K < 128) or ops.use_winograd(tuple(w.shape), s, p, x.shape, True))
    big = Ho * Ho >= hnn._bn_dy_min_hw()                       # the shipped threshold of the arithmetic in use (784 px on fp32 MFMA, 196 on bf16x3)
    dyl = ops.LazyGrad(dy, torch.randn_like(y), torch.randn(4, K, device=dev).contiguous()) if (big and C != 3 and ops.can_lazy_dy(w.shape, s, p)) else None
    sum_in = "conv1" in name and name != "p64.0.conv1" and H * H >= hnn._closing_hw()[0] and ops.can_form_closing_sum(w.shape, s, p)
    if C == 3:
        # which stem kernels the library takes for this shape: asked of the library itself (rows-in-LDS from 224-pixel image rows, else the row-taps gather)
        import ctypes
        dsc = ops.conv_desc(x.shape, tuple(w.shape), s, p)
        lib = _lib.load()
        fv = "rows-in-LDS" if int(lib.ssv_stem_conv_fwd_stats_rows_per_group(ctypes.byref(dsc))) != 64 else "row-taps"
        wv = "rows-in-LDS" if int(lib.ssv_stem_conv_wgrad_rows_per_group(ctypes.byref(dsc))) > 0 else "row-taps"
        wrows = ops.stem_weight_rows(w)
        t_f = timeit(lambda: ops.stem_conv_fwd(x, wrows, tuple(w.shape), s, p, want_stats=True))
        t_w = timeit(lambda: ops.stem_conv_wgrad(x, dy, tuple(w.shape), s, p))
    elif sum_in:
        fv = "stats+sum_in"
        res = torch.randn_like(x)
        t_f = timeit(lambda: ops.conv2d_fwd_sumin(x, res, aff[0], aff[1], None, w, want_mask=True))
        wv = "plain" + ("+dy_in" if dyl is not None else "")
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dyl if dyl is not None else dy, w, dw, s, p, accumulate=True))
        del res
    elif lazy_in:
        wino = ops.use_winograd(tuple(w.shape), s, p, x.shape, True)        # stride-1 3x3 layers of the deep stages: Winograd F(2x2, 3x3)
        form = wino_form(lambda: ops.conv2d_fwd_fused(x, w, s, p, in_affine=aff, want_stats=True, keep_v=True))
        assert bool(form) == bool(wino)
        fv = "stats+bn_in" + (f"+winograd {form}" if form else "")
        t_f = timeit(lambda: ops.conv2d_fwd_fused(x, w, s, p, in_affine=aff, want_stats=True, keep_v=True))
        vkeep = getattr(ops.conv2d_fwd_fused(x, w, s, p, in_affine=aff, want_stats=True, keep_v=True)[0], "_wino_v", None)
        form = wino_form(lambda: ops.conv2d_wgrad(x, dyl if dyl is not None else dy, w, dw, s, p, accumulate=True, in_affine=aff, wino_v=vkeep))
        wv = "bn_in" + ("+dy_in" if dyl is not None else "") + (f"+winograd {form}" if form else "")
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dyl if dyl is not None else dy, w, dw, s, p, accumulate=True, in_affine=aff, wino_v=vkeep))
        del vkeep
    else:
        fv = "stats" if stats_ok else ("c4" if Cx == 4 else "plain")
        t_f = timeit((lambda: ops.conv2d_fwd_stats(x, w, s, p)) if stats_ok else (lambda: ops.conv2d_fwd(x, w, s, p)))
        wv = "plain" + ("+dy_in" if dyl is not None else "")
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dyl if dyl is not None else dy, w, dw, s, p, accumulate=True))
    if C == 3:
        dv, t_d, dwgs = "-", float("nan"), 0
    else:
        # the data gradient feeds a BatchNorm backward: gated epilogue (recomputed gate for the fused chain, byte mask for a unit output)
        gx = torch.randn_like(x)
        mean, invstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        if lazy_in:
            gate = ops.BnGateCtx(gx, mean, invstd, scale=aff[0], shift=aff[1])
        else:
            gate = ops.BnGateCtx(gx, mean, invstd, mask=torch.randint(0, 16, (x.numel() // 4,), device=dev, dtype=torch.uint8))
        addend = torch.randn_like(x) if kind == "plain" else None
        form = wino_form(lambda: ops.conv2d_dgrad(dyl if dyl is not None else dy, w, x.shape, s, p, addend=addend, gate=gate))
        dv = ((f"winograd {form}" if form else ("fwd-kernel" if s == 1 else "dgrad-kernel")) + "+gate" + ("+addend" if addend is not None else "") +
              ("+dy_in" if dyl is not None else ""))
        t_d = timeit(lambda: ops.conv2d_dgrad(dyl if dyl is not None else dy, w, x.shape, s, p, addend=addend, gate=gate))
        if ".1.conv1" in name and kind == "plain" and s == 1:
            # the first of these units sits behind a projection shortcut: its gate also reduces against that BatchNorm's input (GATE 3)
            gate2 = ops.BnGateCtx(gx, mean, invstd, mask=gate.mask, second=(torch.randn_like(x), mean, invstd))
            t_d2 = timeit(lambda: ops.conv2d_dgrad(dyl if dyl is not None else dy, w, x.shape, s, p, addend=addend, gate=gate2))
            print(f"   {name}: data gradient with the second reduction target (1 of its {cnt} instances): {t_d2:.3f} ms, {flop / (t_d2 * 1e-3) / 1e12:.1f} TFLOP/s, "
                  f"{(xin_b(B, H, Cx) * 4 + m * K * 4 * (2 if dyl is not None else 1)) / (t_d2 * 1e-3) / 1e12:.2f} TB/s", flush=True)
            del gate2
        dwgs = grid_fwd(B * H * H, C) if s == 1 else s * s * grid_fwd(B * (-(-H // s)) ** 2, C)
    # algorithmic HBM bytes of each launch in the variant the step uses: every operand stream once (weights and per-channel vectors ignored)
    n_in, n_out = B * H * H, m                                   # input pixels, output pixels
    xin, yout = 4.0 * n_in * Cx, 4.0 * n_out * K
    by_f = xin + yout + ((xin + xin + n_in * Cx / 4) if sum_in else 0)           # sum_in: + shortcut read, activation + byte mask written
    by_w = xin + yout + (yout if dyl is not None else 0)                         # dy_in: + the BatchNorm input beside g
    if C == 3:
        by_d = 0.0
    else:
        by_d = yout + (yout if dyl is not None else 0) + xin                     # dy (+ BatchNorm input), dx written
        by_d += xin if addend is not None else 0                                 # residual gradient read
        by_d += xin + (n_in * Cx / 4 if not lazy_in else 0)                      # gate: the BatchNorm input (+ byte mask)
    tf = lambda t: flop / (t * 1e-3) / 1e12
    peaks = [pipe_peak(x.shape, w.shape, s, p, q) for q in range(3)]
    wgs = grid_fwd(m, K)
    row = [name, cnt, f"{H}x{H}", C, K, f"{R}x{R}", s, round(flop / 1e9, 1), fv, round(t_f, 3), round(tf(t_f), 1), wgs, round(wgs / 768, 2),
           dv, round(t_d, 3), round(tf(t_d), 1) if t_d == t_d else "", dwgs, wv, round(t_w, 3), round(tf(t_w), 1),
           *roof(flop, by_f, t_f, peaks[0]), *roof(flop, by_d, t_d, peaks[1]), *roof(flop, by_w, t_w, peaks[2])]
    rows.append(row)
    print(" ".join(f"{str(v):>13s}" for v in row), flush=True)
    for k, t, nb, pk in (("fwd", t_f, by_f, peaks[0]), ("dgrad", t_d, by_d, peaks[1]), ("wgrad", t_w, by_w, peaks[2])):
        if t == t:
            a = tot.setdefault(k, [0.0, 0.0, 0.0, 0.0])
            a[0] += t * cnt
            a[1] += flop * cnt
            a[2] += nb * cnt
            a[3] += max(flop / pk, nb / HBM_PEAK) * 1e3 * cnt
    del x, y, dy, w, dw
for k, (t, f, nb, bound) in tot.items():
    print(f"total {k:6s}: {t:8.2f} ms per view  {f / (t * 1e-3) / 1e12:6.1f} TFLOP/s  {nb / 1e9:7.2f} GB of operand streams  {nb / (t * 1e-3) / 1e12:5.2f} TB/s  "
          f"sum of the layers' own bounds {bound:6.2f} ms ({bound / t:.3f})")
    # TOTAL rows: GFLOP | ms | TFLOP/s in the fwd columns, then GB of operand streams | TB/s | sum of per-layer bounds / time in the fwd_GB.. columns
    rows.append([f"TOTAL {k}", "", "", "", "", "", "", round(f / 1e9, 1), "", round(t, 2), round(f / (t * 1e-3) / 1e12, 1)] + [""] * 9 +
                [round(nb / 1e9, 2), round(nb / (t * 1e-3) / 1e12, 2), round(bound / t, 3)] + [""] * 6)
if OUT:
    with open(OUT, "w", newline="") as fh:
        wtr = csv.writer(fh)
        wtr.writerow(hdr)
        wtr.writerows(rows)
        wtr.writerow([f"BUILD src_sha16={_lib.source_sha16()} lib_sha16={_lib.lib_sha16()}", f"batch {B}"])      # which build these rows were measured on (bench.py checks it)
