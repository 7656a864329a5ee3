#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient agreement of ONE SimCLR step (HIP vs the fp32 CPU oracle) at a chosen network / size / batch,
for the fused and the unfused BatchNorm paths.   python tools/diag_step_grads.py resnet50 0 224 32"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
from conftest import seeded_randn  # noqa: E402
from test_gpu_step import _Step, rel_l2  # noqa: E402
from ssv_amd import nn as hnn  # noqa: E402

arch, rbc, size, b = sys.argv[1], bool(int(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
torch.set_num_threads(32)
dev = torch.device("cuda:0")
a1, a2 = seeded_randn(1, b, 3, size, size), seeded_randn(2, b, 3, size, size)
o = oracle.SimCLROracle(arch, rbc, 128, lr=0.002, weight_decay=1e-4)
ref = o.train_step(a1, a2)
names = [k for k in o.state() if k.endswith(".weight") or k.endswith(".bias")]
for apply_, bwd_ in ((True, True),):
    hnn._FUSE_BN_APPLY, hnn._FUSE_BN_BWD = apply_, bwd_
    m = _Step(dev, arch, rbc, lr=0.02)
    loss, _, _ = m.step(a1, a2)
    errs = []
    for p, go, off, name in zip(m.params(), o.last_grads, m.optim.arena.offsets, names):
        got = m.grads[off:off + p.numel()]
        got = got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)
        if float(go.norm()) > 1e-5:
            errs.append((rel_l2(got, go), name))
    tail = []
    for p, go, off, name in list(zip(m.params(), o.last_grads, m.optim.arena.offsets, names))[-14:]:
        got = m.grads[off:off + p.numel()]
        got = (got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)).cpu().double()
        gd = go.double()
        tail.append(f"{name}: rel {rel_l2(got, go):.1e} scale {float((got * gd).sum() / (gd * gd).sum()):.5f} |g| {float(gd.norm()):.2e}")
    print("\n".join(tail))
    errs.sort(reverse=True)
    print(f"fuse_apply={apply_} fuse_bwd={bwd_}: loss {loss:.7f} (cpu {ref['loss']:.7f}); worst grads: " + ", ".join(f"{n} {e:.1e}" for e, n in errs[:6]) + f"; median {errs[len(errs) // 2][0]:.1e}", flush=True)
