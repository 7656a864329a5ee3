#!/usr/bin/env python3
"""Do a bandwidth-bound and an MFMA-bound convolution overlap when launched on two HIP streams?
A = 1x1 256->64 on 56x56 (batch 512; ~5 TB/s, 50 TFLOP/s alone), B = 1x1 1024->512 on 14x14 (batch 512; ~130 TFLOP/s alone).
Times REP launches of each: alone, back to back on one stream, and side by side on two streams (A|B, A|A, B|B).
    python tools/probe_overlap.py [REP = 20]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def nhwc(n, h, w, c):
    return torch.randn(n, h, w, c, device=dev, generator=g)


def filt(k, c, r):
    return torch.randn(k, r, r, c, device=dev, generator=g).permute(0, 3, 1, 2)


def case(n, hw, c, k, r):
    return [(nhwc(n, hw, hw, c), filt(k, c, r), 1, r // 2) for _ in range(2)]      # one operand set per stream: no shared cache lines


CASES = {
    "A  1x1 256->64  56x56 b512": case(512, 56, 256, 64, 1),
    "A' 1x1 64->256  56x56 b512": case(512, 56, 64, 256, 1),
    "B  1x1 1024->512 14x14 b512": case(512, 14, 1024, 512, 1),
    "B' 3x3 512->512  7x7 b512 direct": case(512, 7, 512, 512, 3),
}
ops.WINOGRAD = False
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(case, stream, n=REP):
    x, w, st, pd = CASES[case][0 if stream is s1 else 1]
    with torch.cuda.stream(stream):
        for _ in range(n):
            ops.conv2d_fwd(x, w, st, pd)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_event(e0)
    s2.wait_event(e0)
    fn()
    a, b = torch.cuda.Event(), torch.cuda.Event()
    a.record(s1)
    b.record(s2)
    torch.cuda.current_stream().wait_event(a)
    torch.cuda.current_stream().wait_event(b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


alone = {}
for c in CASES:
    alone[c] = timed(lambda: run(c, s1))
    print(f"alone  {c:36s} {alone[c]:7.3f} ms / launch", flush=True)
names = list(CASES)
for i, a in enumerate(names):
    for b in names[i:]:
        t = timed(lambda: (run(a, s1), run(b, s2)))
        print(f"pair   {a[:2]} | {b[:2]}: {t:7.3f} ms per pair   sum alone {alone[a] + alone[b]:7.3f}   max alone {max(alone[a], alone[b]):7.3f}", flush=True)
