#!/usr/bin/env python3
"""Probe (round 5): what the backward of a Winograd F(4x4) layer spends on its output gradient - (a) BatchNorm backward apply pass (writes dy) + the data gradient's input
transform + the weight gradient's dY transform, against (b) ONE ssv_wino44_dy_transform_both pass, with dy materialised and formed on load from (g, x, coefficients).
Also which of the forms one real training step takes (ops.DISPATCH).     python tools/probe/wino44_dy_both_probe.py [batch = 512]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssv_amd import _lib, ops  # noqa: E402

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


for H, K in ((28, 128), (14, 256), (7, 512)):
    g, x = torch.randn(B, H, H, K, device=dev), torch.randn(B, H, H, K, device=dev)
    gamma, mean, invstd = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5
    groups = B * H * H // 64
    part = (torch.randn(groups, K, device=dev), torch.randn(groups, K, device=dev), groups)
    dgam, dbet = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    t = int(lib.ssv_wino44_tiles(B, H, H))
    vd, dm = torch.empty((36, t, K), device=dev), torch.empty((36, t, K), device=dev)
    t_apply = timeit(lambda: ops.bn_bwd_from_partials(g, x, gamma, mean, invstd, part, dgam, dbet))
    dy = ops.bn_bwd_from_partials(g, x, gamma, mean, invstd, part, dgam, dbet)
    t_in = timeit(lambda: _lib.call("ssv_wino44_input_transform", B, H, H, K, _lib.ptr(dy), None, None, _lib.ptr(vd), None, _lib.stream()))
    t_dy = timeit(lambda: _lib.call("ssv_wino44_dy_transform", B, H, H, K, _lib.ptr(dy), _lib.ptr(dm), _lib.stream()))
    t_both = timeit(lambda: _lib.call("ssv_wino44_dy_transform_both", B, H, H, K, _lib.ptr(dy), None, _lib.ptr(vd), _lib.ptr(dm), _lib.stream()))
    coef = ops.bn_bwd_coef(x, gamma, mean, invstd, part, dgam, dbet)
    t_coef = timeit(lambda: ops.bn_bwd_coef(x, gamma, mean, invstd, part, dgam, dbet))
    dyin = _lib.BnDyin(_lib.ptr(x), _lib.ptr(coef))
    t_lazy = timeit(lambda: _lib.call("ssv_wino44_dy_transform_both", B, H, H, K, _lib.ptr(g), ctypes.byref(dyin), _lib.ptr(vd), _lib.ptr(dm), _lib.stream()))
    el = B * H * H * K * 4 / 1e9
    print(f"{H}x{H}x{K}: apply (finalize + element-wise) {t_apply:.3f} | input transform {t_in:.3f} | dY transform {t_dy:.3f}  = {t_apply + t_in + t_dy:.3f} ms   ->   "
          f"both from dy {t_both:.3f} (+ apply {t_apply:.3f} = {t_both + t_apply:.3f})   |   coefficients {t_coef:.3f} + both formed on load {t_lazy:.3f} = {t_coef + t_lazy:.3f} ms"
          f"   [{el:.3f} GB per tensor: formed-on-load pass moves {6.5 * el:.2f} GB -> {6.5 * el / t_lazy:.2f} TB/s]", flush=True)
