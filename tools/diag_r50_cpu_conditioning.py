#!/usr/bin/env python3
"""Diagnostic (CPU only, no GPU): how far is the fp32 CPU oracle from its own fp64 twin on a ResNet-50 224x224 SimCLR trajectory?
    python tools/diag_r50_cpu_conditioning.py <batch> <lr after seeding> <steps> [same|fresh] [corr|noise]
Answers "at which learning rate / batch is a per-step 1e-4 comparison of two fp32 evaluations well posed at the BENCH shape" without
spending GPU time: prints, per step, cpu32, cpu64, their relative distance, and how far the step moved the loss (fp64)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
from conftest import seeded_randn  # noqa: E402
from test_gpu_step import _oracle64_like  # noqa: E402


def views(seed, b, kind, size=224):
    if kind == "corr":      # two noisy views of one smooth image per sample
        base = torch.nn.functional.interpolate(seeded_randn(seed, b, 3, 7, 7), size=size, mode="bilinear", align_corners=False) * 2.0
        return base + 0.3 * seeded_randn(seed + 1, b, 3, size, size), base + 0.3 * seeded_randn(seed + 2, b, 3, size, size)
    return seeded_randn(seed, b, 3, size, size), seeded_randn(seed + 1, b, 3, size, size)


def main():
    b, lr, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    same = (sys.argv[4] if len(sys.argv) > 4 else "fresh") == "same"
    kind = sys.argv[5] if len(sys.argv) > 5 else "noise"
    make = lambda: oracle.SimCLROracle("resnet50", False, 128, lr=lr, weight_decay=1e-4)
    o32, o64 = make(), _oracle64_like(make)
    prev = None
    for s in range(steps):
        a1, a2 = views(5000 + (0 if same else 3 * s), b, kind)
        t0 = time.time()
        r32 = o32.train_step(a1, a2, return_z=True)
        t1 = time.time()
        r64 = o64.train_step(a1.double(), a2.double(), return_z=True)
        t2 = time.time()
        dz = float((r32["z_1"].double() - r64["z_1"]).abs().max())
        g32, g64 = o32.last_grads, o64.last_grads
        errs = sorted(float((a.double() - c).norm() / (c.norm() + 1e-30)) for a, c in zip(g32, g64) if float(c.norm()) > 1e-5)
        gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g64)))
        wn = float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in o64.params)))
        moved = "" if prev is None else f" moved {abs(r64['loss'] - prev) / abs(prev):.2e}"
        prev = r64["loss"]
        print(f"step {s}: cpu32 {r32['loss']:.7f} cpu64 {r64['loss']:.7f} |32-64| {abs(r32['loss'] - r64['loss']) / abs(r64['loss']):.2e}{moved}  max|dz| {dz:.2e}  "
              f"grad err median {errs[len(errs) // 2]:.2e} worst {errs[-1]:.2e}  |g| {gn:.2e} |w| {wn:.2e}  ({t1 - t0:.0f}s fp32, {t2 - t1:.0f}s fp64)", flush=True)


if __name__ == "__main__":
    main()
