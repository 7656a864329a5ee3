"""r06: the step-graph lifetime without a timing heuristic (ssv_amd/graph.py, "when a graph is DESTROYED"): capture a two-stream training step, replay it, take it out of
service, run the step eagerly - over and over, as tools/exp/r05_graph_event_stress.py did when it found the destroy-under-callbacks race.  A graph that leaves service is
retired; a retired generation is destroyed only after the device went idle twice with a capture in between, so no graph is ever destroyed near its last launch.  Also
prints what an LRU of MAX_LIVE graphs holds in device memory (the private pools of the captured steps).
    python tools/exp/r06_graph_lru_stress.py [cycles = 1000]"""
import gc, os, sys, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.enable()
import torch
import bench
from ssv_amd import graph
from ssv_amd.graph import StepGraph
dev = torch.device("cuda:0")
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator(device=dev).manual_seed(0)
batches = [{"aug_1": torch.randn(32, 3, 32, 32, device=dev, generator=g), "aug_2": torch.randn(32, 3, 32, 32, device=dev, generator=g)} for _ in range(3)]
t0 = time.time()
step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
t = step.trainer
for c in range(cycles):
    if c % 25 == 24:                                   # a fresh trainer now and then: its modules' events go after their last graph
        del step, t
        gc.collect()
        step, _ = bench.build(dev, ("simclr", "byol", "barlow")[(c // 25) % 3], arch="resnet18", reduce_bottom_conv=True)
        t = step.trainer
    sg = StepGraph(t, mode="1", graph_floors=bool(c % 2))
    for i in range(4):                                 # 2 eager steps, the capture + its replay, one more replay
        loss = sg(batches[i % 3])["loss"]
        t._after_step(i)
    sg.close()                                         # out of service: retired, not destroyed
    del sg
    for i in range(2):                                 # ... and the step runs eagerly on the same modules
        loss = t.train_step(batches[i])["loss"]
        t._after_step(i)
    if c % 50 == 0:
        print(c, round(loss, 4), "retired", len(graph._RETIRED[0]) + len(graph._RETIRED[1]), "GB reserved", round(torch.cuda.memory_reserved() / 2**30, 2), round(time.time() - t0, 1), flush=True)
print("done", cycles, round(time.time() - t0, 1), flush=True)
# what an LRU of MAX_LIVE graphs holds: one trainer, MAX_LIVE graph keys (a trainer-side scalar that keys the graph), batch 64 and 512
for bs in (64, 512):
    gc.collect(); graph._reap(final=True); torch.cuda.empty_cache()
    step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
    t = step.trainer
    b = {"aug_1": torch.randn(bs, 3, 32, 32, device=dev, generator=g), "aug_2": torch.randn(bs, 3, 32, 32, device=dev, generator=g)}
    sg = StepGraph(t, mode="1")
    for i in range(3):
        sg(b)
    torch.cuda.synchronize()
    base = torch.cuda.memory_reserved()
    tick = [0]
    t.graph_key = lambda: (tick[0],)
    for k in range(1, graph.MAX_LIVE + 1):
        tick[0] = k
        sg(b); sg(b)
    torch.cuda.synchronize()
    print(f"batch {bs}: {len(sg.graphs)} live graphs hold {(torch.cuda.memory_reserved() - base) / 2**30 / (graph.MAX_LIVE - 1 + 1e-9):.2f} GB each "
          f"({torch.cuda.memory_reserved() / 2**30:.2f} GB reserved in all)", flush=True)
    sg.close(); del sg, step, t
