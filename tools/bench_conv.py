"""Per-layer-shape timing of the conv kernels (ResNet-50 @224, given batch): TFLOP/s for fwd / dgrad / wgrad."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssv_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
shapes = []   # (name, H, W, C, K, R, stride, pad, count)
shapes.append(("stem7x7", 224, 224, 3, 64, 7, 2, 3, 1))
cin, h = 64, 56
for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
    for b in range(blocks):
        s = stride if b == 0 else 1
        shapes.append((f"l{planes}.{b}.c1", h, h, cin, planes, 1, 1, 0, 1))
        shapes.append((f"l{planes}.{b}.c2", h, h, planes, planes, 3, s, 1, 1))
        h2 = (h + 2 - 3) // s + 1
        shapes.append((f"l{planes}.{b}.c3", h2, h2, planes, planes * 4, 1, 1, 0, 1))
        if b == 0:
            shapes.append((f"l{planes}.{b}.ds", h, h, cin, planes * 4, 1, s, 0, 1))
        cin, h = planes * 4, h2
# merge identical shapes
uniq = {}
for n, H, W, C, K, R, s, p, c in shapes:
    key = (H, W, C, K, R, s, p)
    if key in uniq:
        uniq[key][1] += 1
    else:
        uniq[key] = [n, 1]

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP

tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
print(f"B={B}  {'layer':12s} {'HxW':>7s} {'C':>5s} {'K':>5s} R s  cnt | {'fwd ms':>8s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s}")
for (H, W, C, K, R, s, p), (name, cnt) in uniq.items():
    x = torch.randn(B, H, W, C, device=dev)
    w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd(x, w, s, p)
    dy = torch.randn_like(y)
    dw = torch.zeros_like(w)
    flop = 2.0 * y.numel() * C * R * R
    t_f = timeit(lambda: ops.conv2d_fwd(x, w, s, p))
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, s, p, accumulate=True))
    if C % 4 == 0 and K % 16 == 0:
        t_d = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p))
    else:
        t_d = float("nan")
    tf = lambda t: flop / (t * 1e-3) / 1e12
    t_df = float("nan")
    if s == 1 and C % 4 == 0 and K % 16 == 0:      # stride 1: dgrad == forward convolution of dy with the transposed, 180-degree rotated filter
        wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)      # [C, K, R, R] in OHWI memory
        ref = ops.conv2d_dgrad(dy, w, x.shape, s, p)
        alt = ops.conv2d_fwd(dy, wt, 1, R - 1 - p)
        assert alt.shape == ref.shape and float((alt - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-6
        t_df = timeit(lambda: ops.conv2d_fwd(dy, wt, 1, R - 1 - p))
    print(f"      {name:12s} {H:3d}x{W:<3d} {C:5d} {K:5d} {R} {s} {cnt:4d} | {t_f:8.3f} {tf(t_f):6.1f} | {t_d:8.3f} {tf(t_d):6.1f} | {t_w:8.3f} {tf(t_w):6.1f} | dgrad-as-fwd {t_df:8.3f} {tf(t_df):6.1f}")
    for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
        if t == t:
            tot[k][0] += t * cnt; tot[k][1] += flop * cnt
for k, (t, f) in tot.items():
    print(f"total {k:6s}: {t:8.2f} ms  {f / (t * 1e-3) / 1e12:6.1f} TF")
