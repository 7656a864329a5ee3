#!/usr/bin/env python3
"""Step time of the reference's own CIFAR configuration (configs/simclr.yaml: resnet18 reduce_bottom_conv, 32x32, bs 512) and how much of
it is host launch overhead: ms/step synchronised every step (the reference's loss.item()), and the host's pure enqueue time.
    python tools/bench_cifar.py [batch = 512]                 the JSON line
    python tools/bench_cifar.py <batch> steps <n>             n steps after 3 warm-up steps, nothing printed but one line: the program rocprofv3 is given"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssv_amd import nn as hnn
from ssv_amd.models import heads
from ssv_amd.networks import resnet
from ssv_amd.utils import losses, train_utils

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(420)
enc = resnet.resnet18(reduce_bottom_conv=True).to(dev)
head = heads.SimclrProjectionHead(512, 128).to(dev)
opt = train_utils.get_optimizer({"name": "sgd", "lr": 0.2, "weight_decay": 1e-4}, list(enc.parameters()) + list(head.parameters()))
loss_fn = losses.SimclrLoss(True, 0.5)
a1, a2 = torch.randn(bs, 3, 32, 32, device=dev), torch.randn(bs, 3, 32, 32, device=dev)


def step(sync):
    with hnn.parallel_views(dev) as pv:
        with pv.view(0):
            z1 = head(enc(a1))
        with pv.view(1):
            z2 = head(enc(a2))
    loss = loss_fn(z1, z2)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.item() if sync else None


if len(sys.argv) > 3 and sys.argv[2] == "steps":
    for _ in range(3):
        step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(int(sys.argv[3])):
        step(True)
    torch.cuda.synchronize()
    print(json.dumps({"workload": f"SimCLR resnet18 (reduce_bottom_conv) 32x32 bs {bs}", "steps": int(sys.argv[3]), "ms_per_step": round((time.perf_counter() - t0) / int(sys.argv[3]) * 1e3, 2)}))
    sys.exit(0)
for _ in range(5):
    step(True)
out = {}
for sync in (True, False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(sync)
    t_host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 20
    out["sync_every_step" if sync else "free_running"] = {"host_enqueue_ms": round(t_host * 1e3, 2), "ms_per_step": round(t_all * 1e3, 2),
                                                          "images_per_s": round(bs / t_all, 1)}
# algorithmic work of the step: 1.679 GFLOP / sample (0.840 per view: conv fwd 0.1402 GMAC + bwd, projector) -> fraction of the fp32 MFMA roof
gflop = 1.679 * bs
out["free_running"]["whole_step_mfma_frac"] = round(gflop / out["free_running"]["ms_per_step"] / 157.3, 4)
print(json.dumps({"workload": f"SimCLR resnet18 (reduce_bottom_conv) 32x32 bs {bs}", "algorithmic_gflop_per_step": round(gflop, 1), **out}))
