#!/usr/bin/env python3
"""Diagnostic: per-step losses of the HIP path, the fp32 CPU oracle and an fp64 evaluation of the oracle on the same inputs.
    python tools/diag_trajectory.py simclr|barlow|byol <config lr> <batch> <steps>
Prints one line per step: hip, cpu32, cpu64 and the two distances to fp64 (relative)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
from conftest import seeded_randn  # noqa: E402
from test_gpu_step import _Step, _bare_trainer, _oracle64_like  # noqa: E402


def main():
    algo, lr, b, steps = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda:0")
    if algo == "simclr":
        m = _Step(dev, "resnet18", True, lr=lr)
        eff = m.optim.param_groups[0]["lr"]
        make = lambda: oracle.SimCLROracle("resnet18", True, 128, lr=eff, weight_decay=1e-4)
        hip = lambda a1, a2, s: m.step(a1, a2)[0]
        parts = lambda o: (o.encoder, o.proj_head)
    elif algo == "barlow":
        from ssv_amd.models.barlow import BarlowTwins
        cfg = {"epochs": 1000, "proj_dim": 256, "encoder": {"reduce_bottom_conv": True}, "optimizer": {"name": "sgd", "lr": lr, "weight_decay": 1.5e-6},
               "scheduler": {"name": "cosine", "warmup_epochs": 10}, "loss_fn": {"normalize": False, "off_diagonal_weight": 0.005}}
        t = _bare_trainer(BarlowTwins, dev, cfg)
        eff = t.optim.param_groups[0]["lr"]
        make = lambda: oracle.BarlowOracle("resnet18", True, 256, lr=eff, weight_decay=1.5e-6, normalize=False)
        hip = lambda a1, a2, s: t.train_step({"aug_1": a1, "aug_2": a2})["loss"]
    else:
        raise SystemExit("simclr | barlow")
    o32, o64 = make(), _oracle64_like(make)
    for s in range(steps):
        if len(sys.argv) > 5 and sys.argv[5] == "corr":      # two noisy views of one smooth image per sample (what the algorithms are built for)
            base = torch.nn.functional.interpolate(seeded_randn(2100 + 3 * s, b, 3, 4, 4), size=32, mode="bilinear", align_corners=False) * 2.0
            a1, a2 = base + 0.3 * seeded_randn(2101 + 3 * s, b, 3, 32, 32), base + 0.3 * seeded_randn(2102 + 3 * s, b, 3, 32, 32)
        else:
            a1, a2 = seeded_randn(2100 + 2 * s, b, 3, 32, 32), seeded_randn(2101 + 2 * s, b, 3, 32, 32)
        h = hip(a1, a2, s)
        c32 = o32.train_step(a1, a2)["loss"]
        c64 = o64.train_step(a1.double(), a2.double())["loss"]
        print(f"step {s}: hip {h:.7f} cpu32 {c32:.7f} cpu64 {c64:.7f}  |hip-64| {abs(h - c64) / abs(c64):.2e}  |32-64| {abs(c32 - c64) / abs(c64):.2e}  |hip-32| {abs(h - c32) / abs(c32):.2e}", flush=True)


if __name__ == "__main__":
    main()
