"""r05: stress of the step graph in the production pattern - ONE trainer, a re-capture every `period` steps (the schedule moves the learning rate once per epoch), eager
feature extraction between epochs; and in the test pattern - many trainers one after the other.   python tools/exp/r05_graph_stress.py prod|many [iterations]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ssv_amd.graph import StepGraph
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "prod"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = torch.Generator(device=dev).manual_seed(0)
batches = [{"aug_1": torch.randn(64, 3, 32, 32, device=dev, generator=g), "aug_2": torch.randn(64, 3, 32, 32, device=dev, generator=g)} for _ in range(4)]
probe = torch.randn(64, 3, 32, 32, device=dev, generator=g)
t0 = time.time()
if mode == "prod":
    step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
    t = step.trainer
    for ep in range(iters):
        for grp in t.optim.param_groups:
            grp["lr"] = 0.2 * (1.0 - 0.5 * ep / iters)
        for i in range(12):
            loss = t.step(batches[i % 4])["loss"]
        with torch.no_grad():
            t._features(probe)
        if ep % 20 == 0:
            print(ep, round(loss, 4), t._step_graph.describe(), round(time.time() - t0, 1), flush=True)
else:
    for k in range(iters):
        step, _ = bench.build(dev, ("simclr", "byol", "barlow")[k % 3], arch="resnet18", reduce_bottom_conv=True)
        t = step.trainer
        sg = StepGraph(t, mode="1", graph_floors=bool(k % 2))
        for i in range(8):
            if i == 5:
                for grp in t.optim.param_groups:
                    grp["lr"] *= 0.5
            loss = sg(batches[i % 4])["loss"]
            t._after_step(i)
        if k % 10 == 0:
            print(k, round(loss, 4), sg.describe(), round(time.time() - t0, 1), flush=True)
print("done", round(time.time() - t0, 1))
