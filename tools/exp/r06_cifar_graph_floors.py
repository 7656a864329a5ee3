"""r06: the graph-mode Winograd floors (ops.graph_dispatch: channels >= 64, F(2x2) tiles >= 64, F(4x4) tiles >= 256, set in round 5 on the fp32 arithmetic) re-measured
on the bf16x3 arithmetic: config 1's step (ResNet-18, 32 x 32) replayed as one HIP graph, ms per step at batch 64 (and 512) per floors triple, two rounds.
    python tools/exp/r06_cifar_graph_floors.py [batch = 64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ssv_amd import ops
from ssv_amd.graph import StepGraph

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = {"aug_1": torch.randn(bs, 3, 32, 32, device=dev), "aug_2": torch.randn(bs, 3, 32, 32, device=dev)}
FLOORS = [(64, 64, 256), (64, 64, 64), (64, 16, 64), (64, 16, 256), (128, 64, 256), (64, 256, 1024), (64, 64, 10 ** 9), (10 ** 9, 64, 256)]
for rnd in range(2):
    for ch, t2, t4 in FLOORS:
        ops.graph_dispatch.CHANNELS, ops.graph_dispatch.TILES, ops.graph_dispatch.TILES44 = ch, t2, t4
        train_step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
        sg = StepGraph(train_step.trainer, mode="1")
        for _ in range(4):
            sg(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            sg(batch)
        torch.cuda.synchronize()
        print(f"bs {bs} channels >= {ch:<10d} F(2x2) tiles >= {t2:<5d} F(4x4) tiles >= {t4:<10d} {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step", flush=True)
        sg.close()
        del sg, train_step
