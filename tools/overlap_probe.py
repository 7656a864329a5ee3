#!/usr/bin/env python3
"""Do the REAL kernels overlap?  One stream loops a convolution (forward with the statistics epilogue, ResNet-50 layer shapes), a second
stream loops a BatchNorm element-wise pass (ssv_bn_apply with a residual: two reads + one write per element) on an unrelated tensor.
Prints each alone and both together: with perfect overlap `together` = max(alone), with none = sum.

    [SSV_HIP_LIB=...] python tools/overlap_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 512
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn_a, na, fn_b, nb):
    torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    s1.wait_event(e0)
    s2.wait_event(e0)
    if fn_a:
        with torch.cuda.stream(s1):
            for _ in range(na):
                fn_a()
            ea.record()
    if fn_b:
        with torch.cuda.stream(s2):
            for _ in range(nb):
                fn_b()
            eb.record()
    torch.cuda.synchronize()
    ta = e0.elapsed_time(ea) if fn_a else 0.0
    tb = e0.elapsed_time(eb) if fn_b else 0.0
    return ta, tb


def main():
    cases = [("3x3 256->256 @14", 14, 256, 256, 3), ("1x1 1024->256 @14", 14, 1024, 256, 1), ("3x3 128->128 @28", 28, 128, 128, 3)]
    m, c = B * 28 * 28, 512                                   # the BatchNorm tensor: layer2 block output
    with torch.cuda.stream(s2):
        bx = torch.randn(m, c, device=dev)
        br = torch.randn(m, c, device=dev)
        sc = torch.rand(c, device=dev) + 0.5
        sh = torch.randn(c, device=dev)
    bn = lambda: ops.bn_apply(bx.view(B, 28, 28, c), sc, sh, relu=True, residual=br.view(B, 28, 28, c), want_mask=True)
    for name, h, cin, cout, r in cases:
        with torch.cuda.stream(s1):
            x = torch.randn(B, h, h, cin, device=dev)
            w = (torch.randn(cout, cin, r, r, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        conv = lambda: ops.conv2d_fwd_stats(x, w, 1, r // 2)
        torch.cuda.synchronize()
        for f, st in ((conv, s1), (bn, s2)):
            with torch.cuda.stream(st):
                for _ in range(3):
                    f()
        na, nb = 40, 20
        ta, _ = timed(conv, na, None, 0)
        _, tb = timed(None, 0, bn, nb)
        ta2, tb2 = timed(conv, na, bn, nb)
        print(f"{name}: conv x{na} alone {ta:.2f} ms, bn_apply x{nb} alone {tb:.2f} ms | together conv {ta2:.2f}, bn {tb2:.2f} -> wall {max(ta2, tb2):.2f} vs sum {ta + tb:.2f}", flush=True)


if __name__ == "__main__":
    main()
