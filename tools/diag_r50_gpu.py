#!/usr/bin/env python3
"""Diagnostic (GPU box): SimCLR ResNet-50 at 224x224 - the HIP path, the fp32 CPU oracle and its fp64 twin on the same inputs.
    python tools/diag_r50_gpu.py <batch> <lr after seeding> <steps> [same|fresh] [corr|noise] [threads]
Per step: the three losses and their relative distances, max |dz|; on step 0 also the per-tensor gradient errors against fp64 (HIP and CPU
fp32: median / worst / best) - the numbers the r50-224 parity tests' tolerances come from."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import oracle  # noqa: E402
from test_gpu_step import _Step, _oracle64_like, rel_l2  # noqa: E402
from diag_r50_cpu_conditioning import views  # noqa: E402


def main():
    b, lr, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    same = (sys.argv[4] if len(sys.argv) > 4 else "fresh") == "same"
    kind = sys.argv[5] if len(sys.argv) > 5 else "corr"
    torch.set_num_threads(int(sys.argv[6]) if len(sys.argv) > 6 else 32)
    dev = torch.device("cuda:0")
    m = _Step(dev, "resnet50", False, lr=lr * 10)                 # get_scheduler seeds lr / warm-up epochs (10)
    eff = m.optim.param_groups[0]["lr"]
    make = lambda: oracle.SimCLROracle("resnet50", False, 128, lr=eff, weight_decay=1e-4)
    o32, o64 = make(), _oracle64_like(make)
    names = [k for k in o32.state() if k.endswith(".weight") or k.endswith(".bias")]
    first64 = None
    for s in range(steps):
        a1, a2 = views(5000 + (0 if same else 3 * s), b, kind)
        hip, z1, _ = m.step(a1, a2, dual=True)
        t0 = time.time()
        r32 = o32.train_step(a1, a2, return_z=True)
        t1 = time.time()
        r64 = o64.train_step(a1.double(), a2.double(), return_z=True)
        t2 = time.time()
        c32, c64 = r32["loss"], r64["loss"]
        ez_h = float((z1.cpu().double() - r64["z_1"]).abs().max())
        ez_c = float((r32["z_1"].double() - r64["z_1"]).abs().max())
        moved = "" if (first64 is None or not same) else f" moved {abs(c64 - first64) / abs(first64):.2e}"
        first64 = c64
        print(f"step {s}: hip {hip:.7f} cpu32 {c32:.7f} cpu64 {c64:.7f}  |hip-64| {abs(hip - c64) / abs(c64):.2e} |32-64| {abs(c32 - c64) / abs(c64):.2e} "
              f"|hip-32| {abs(hip - c32) / abs(c32):.2e}{moved}  dz hip {ez_h:.2e} cpu {ez_c:.2e}  ({t1 - t0:.0f}s fp32, {t2 - t1:.0f}s fp64)", flush=True)
        if s == 0:
            eh, ec, rows = [], [], []
            for p, g32, g64, off, name in zip(m.params(), o32.last_grads, o64.last_grads, m.optim.arena.offsets, names):
                if float(g64.norm()) < 1e-5:
                    continue
                got = m.grads[off:off + p.numel()]
                got = got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)
                eh.append(rel_l2(got, g64))
                ec.append(rel_l2(g32, g64))
                rows.append((eh[-1], ec[-1], name))
            print(f"   gradients vs fp64, per tensor: hip median {np.median(eh):.2e} worst {max(eh):.2e} best {min(eh):.2e} | cpu32 median {np.median(ec):.2e} worst {max(ec):.2e} best {min(ec):.2e}")
            for e_h, e_c, name in rows[-8:]:
                print(f"     {name:40s} hip {e_h:.2e} cpu {e_c:.2e}")
            for e_h, e_c, name in rows[:4]:
                print(f"     {name:40s} hip {e_h:.2e} cpu {e_c:.2e}")


if __name__ == "__main__":
    main()
