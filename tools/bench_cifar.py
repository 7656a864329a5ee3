#!/usr/bin/env python3
"""Step time of the reference's own CIFAR configuration (configs/simclr.yaml: resnet18 reduce_bottom_conv, 32x32, bs 512) and how much of
it is host launch overhead: ms/step synchronised every step (the reference's loss.item()), and the host's pure enqueue time.
    python tools/bench_cifar.py [batch = 512]                 the JSON line
    python tools/bench_cifar.py <batch> steps <n>             n eager steps after 4 warm-up steps, one line: the program rocprofv3 is given
    python tools/bench_cifar.py <batch> graph_steps <n>       the same through the step graph"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
import bench  # noqa: E402  (the repo's bench.py: build() constructs the package's own SimCLR trainer the way its __init__ does)
from ssv_amd.graph import StepGraph  # noqa: E402

train_step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
trainer = train_step.trainer
batch = {"aug_1": torch.randn(bs, 3, 32, 32, device=dev), "aug_2": torch.randn(bs, 3, 32, 32, device=dev)}


def step(_sync=True):
    return trainer.train_step(batch)["loss"]


if len(sys.argv) > 3 and sys.argv[2] in ("steps", "graph_steps"):
    sg = StepGraph(trainer, mode="1" if sys.argv[2] == "graph_steps" else "0")
    for _ in range(4):
        sg(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(int(sys.argv[3])):
        sg(batch)
    torch.cuda.synchronize()
    print(json.dumps({"workload": f"SimCLR resnet18 (reduce_bottom_conv) 32x32 bs {bs}", "steps": int(sys.argv[3]), "graph": sg.describe(),
                      "ms_per_step": round((time.perf_counter() - t0) / int(sys.argv[3]) * 1e3, 2)}))
    sys.exit(0)
for _ in range(5):
    step(True)
out = {}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step(True)
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
out["eager"] = {"host_ms": round(t_host * 1e3, 2), "ms_per_step": round(t_all * 1e3, 2), "images_per_s": round(bs / t_all, 1)}
# the same steps replayed as one HIP graph (ssv_amd.graph.StepGraph): a fresh trainer, 4 steps to warm up and capture, then 50 timed replays
train_step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
sg = StepGraph(train_step.trainer, mode="1")
for _ in range(4):
    sg(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    sg(batch)
torch.cuda.synchronize()
t_g = (time.perf_counter() - t0) / 50
out["hip_graph"] = {"ms_per_step": round(t_g * 1e3, 2), "images_per_s": round(bs / t_g, 1), "state": sg.describe()}
# algorithmic work of the step: 1.679 GFLOP / sample (0.840 per view: conv fwd 0.1402 GMAC + bwd, projector) -> fraction of the fp32 MFMA roof
gflop = 1.679 * bs
for k in ("eager", "hip_graph"):
    out[k]["whole_step_mfma_frac"] = round(gflop / out[k]["ms_per_step"] / 157.3, 4)
print(json.dumps({"workload": f"SimCLR resnet18 (reduce_bottom_conv) 32x32 bs {bs}, the loss read every step (models/simclr.py:95)", "algorithmic_gflop_per_step": round(gflop, 1), **out}))
