#!/usr/bin/env python3
"""SimCLR step on `-m resnext50` (ResNeXt-50 32x4d, 224x224): ms/step and images/s with the grouped 3x3 convolutions on group-aware tiles (shipped)
and as plain dense block-diagonal products (SSV_NO_GROUP_AWARE_TILES=1), next to resnet50 at the same batch.
    python tools/bench_resnext.py [batch = 256] [steps = 6]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import nn as hnn  # noqa: E402
from ssv_amd.models import heads  # noqa: E402
from ssv_amd.networks import resnet  # noqa: E402
from ssv_amd.utils import losses, train_utils  # noqa: E402

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def run(arch, aware):
    hnn._GROUP_AWARE = aware
    torch.manual_seed(420)
    enc = getattr(resnet, arch)(reduce_bottom_conv=False).to(dev)
    head = heads.SimclrProjectionHead(2048, 128).to(dev)
    opt = train_utils.get_optimizer({"name": "sgd", "lr": 0.02, "weight_decay": 1e-4}, list(enc.parameters()) + list(head.parameters()))
    loss_fn = losses.SimclrLoss(True, 0.5)
    a1, a2 = torch.randn(bs, 3, 224, 224, device=dev), torch.randn(bs, 3, 224, 224, device=dev)

    def step():
        with hnn.parallel_views(dev) as pv:
            with pv.view(0):
                z1 = head(enc(a1))
            with pv.view(1):
                z2 = head(enc(a2))
        loss = loss_fn(z1, z2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        last = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"arch": arch, "group_aware_tiles": aware, "ms_per_step": round(dt * 1e3, 2), "images_per_s": round(bs / dt, 1), "loss": round(float(last), 6)}
    del enc, head, opt
    torch.cuda.empty_cache()
    return out


for arch, aware in (("resnext50_32x4d", True), ("resnext50_32x4d", False), ("resnet50", True)):
    print(json.dumps({"batch": bs, **run(arch, aware)}), flush=True)
