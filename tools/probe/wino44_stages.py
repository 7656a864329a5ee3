#!/usr/bin/env python3
"""Stage times of the F(4x4, 3x3) chain next to F(2x2)'s: input transform, batched GEMMs, output transform (forward with statistics; data gradient with the gate).
    python tools/probe/wino44_stages.py [batch = 512]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssv_amd import _lib, ops  # noqa: E402
from ssv_amd._lib import call, ptr, stream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, rep=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep


for name, H, Cc in (("56x56x64", 56, 64), ("28x28x128", 28, 128), ("14x14x256", 14, 256), ("7x7x512", 7, 512)):
    x = torch.randn(B, H, H, Cc, device=dev)
    w = (torch.randn(Cc, Cc, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    mean, invstd = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5
    gate = ops.BnGateCtx(x, mean, invstd, scale=sc, shift=sh)
    t2, t4 = int(lib.ssv_wino_tiles(B, H, H)), int(lib.ssv_wino44_tiles(B, H, H))
    wt, wsh = ops._ohwi(w)
    u2, u4 = ops._wino_filter(wt, wsh), ops._wino44_filter(wt, wsh)
    v2 = torch.empty(16, t2, Cc, device=dev); m2 = torch.empty(16, t2, Cc, device=dev)
    v4 = torch.empty(36, t4, Cc, device=dev); m4 = torch.empty(36, t4, Cc, device=dev)
    y = torch.empty(B, H, H, Cc, device=dev)
    g2, g4s, g4g = int(lib.ssv_wino_groups(B, H, H)), int(lib.ssv_wino44_groups(B, H, H, 1)), int(lib.ssv_wino44_groups(B, H, H, 0))
    p2, p4 = torch.empty(2, g2, Cc, device=dev), torch.empty(2, max(g4s, g4g), Cc, device=dev)
    st2, _ = ops._gate_struct(gate, g2, Cc, x)
    st4, _ = ops._gate_struct(gate, g4g, Cc, x)
    stats2 = int(lib.ssv_wino_stats_rows_per_group(B, H, H)) > 0
    r = {
        "F(2x2) in (BN on load)": timeit(lambda: call("ssv_wino_input_transform", B, H, H, Cc, ptr(x), ptr(sc), ptr(sh), ptr(v2), stream())),
        "F(2x2) in (plain)": timeit(lambda: call("ssv_wino_input_transform", B, H, H, Cc, ptr(x), None, None, ptr(v2), stream())),
        "F(2x2) gemm": timeit(lambda: call("ssv_gemm_batched", 16, t2, Cc, Cc, ptr(v2), ptr(u2), ptr(m2), stream())),
        "F(2x2) out (stats)": timeit(lambda: call("ssv_wino_output_transform", B, H, H, Cc, ptr(m2), ptr(y), ptr(p2[0]) if stats2 else None, ptr(p2[1]) if stats2 else None, None, stream())),
        "F(2x2) out (gate)": timeit(lambda: call("ssv_wino_output_transform", B, H, H, Cc, ptr(m2), ptr(y), None, None, C.byref(st2), stream())),
        "F(4x4) in (BN on load, + V2)": timeit(lambda: call("ssv_wino44_input_transform", B, H, H, Cc, ptr(x), ptr(sc), ptr(sh), ptr(v4), ptr(v2), stream())),
        "F(4x4) in (BN on load)": timeit(lambda: call("ssv_wino44_input_transform", B, H, H, Cc, ptr(x), ptr(sc), ptr(sh), ptr(v4), None, stream())),
        "F(4x4) in (plain)": timeit(lambda: call("ssv_wino44_input_transform", B, H, H, Cc, ptr(x), None, None, ptr(v4), None, stream())),
        "F(4x4) gemm": timeit(lambda: call("ssv_gemm_batched", 36, t4, Cc, Cc, ptr(v4), ptr(u4), ptr(m4), stream())),
        "F(4x4) out (stats)": timeit(lambda: call("ssv_wino44_output_transform", B, H, H, Cc, ptr(m4), ptr(y), ptr(p4[0]), ptr(p4[1]), None, stream())),
        "F(4x4) out (gate)": timeit(lambda: call("ssv_wino44_output_transform", B, H, H, Cc, ptr(m4), ptr(y), None, None, C.byref(st4), stream())),
    }
    print(name + ":  " + "  |  ".join(f"{k} {v:.3f}" for k, v in r.items()), flush=True)
