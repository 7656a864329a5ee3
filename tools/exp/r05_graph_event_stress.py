"""r05: the crash behind the GPU suite's three rare deaths, isolated: capture a two-stream training step, replay it, DESTROY the graph, run the step eagerly - over and over.
With SSV_GRAPH_DESTROY_GRACE=0 (no wait between the device going idle and the graph's destruction) the host heap is corrupted within ~10-60 cycles ('corrupted size vs.
prev_size', 'malloc(): invalid size', segmentation faults; the one native backtrace caught: libhsa-runtime64's asynchronous handler thread inside libamdhip64 callbacks);
with the default grace of 50 ms: 3 x 300 cycles clean.  STRESS_VARIANT=keep (graphs never destroyed) and =single (one stream) never crashed either; no BatchNorm
ordering events (nobn), no device-to-host loss node (noloss), every capture-time event object kept alive with its graph: still crashed - it is the destruction itself.
    python tools/exp/r05_graph_event_stress.py [cycles = 150]"""
import gc, os, sys, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.enable()
import torch
import bench
from ssv_amd.graph import StepGraph
dev = torch.device("cuda:0")
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 150
VAR = os.environ.get("STRESS_VARIANT", "")          # bisection: keep (graphs never destroyed), single (one stream), noeager (no eager steps after close), nofloors, nogc
from ssv_amd import nn as hnn
if "single" in VAR:
    hnn.set_view_streams(False)
if "nobn" in VAR:                                     # no BatchNorm ordering events at all (racy running statistics: irrelevant here)
    hnn._bn_order_wait = lambda bn, x: None
    hnn._bn_order_record = lambda bn, x: None
if "noloss" in VAR:                                   # no device-to-host copy node in the graph
    _orig_init = hnn.early_item.__init__
    def _init(self, t):
        if hnn._CAPTURE_HOST is not None:
            self.t, self.ev, self.captured = t.detach(), None, True
            return
        _orig_init(self, t)
    hnn.early_item.__init__ = _init
if "nows" in VAR:                                     # the captured step shares the eager workspace (never freed with a graph)
    import ssv_amd.graph as G
    _cap = G.StepGraph._capture
    def _capture(self, batch, ins, key):
        from ssv_amd import ops
        saved = ops.workspace.buf
        class _Keep(dict):
            pass
        rec = _cap(self, batch, ins, key)
        ops.workspace.buf = saved
        return rec
g = torch.Generator(device=dev).manual_seed(0)
batches = [{"aug_1": torch.randn(32, 3, 32, 32, device=dev, generator=g), "aug_2": torch.randn(32, 3, 32, 32, device=dev, generator=g)} for _ in range(3)]
KEEP = []
t0 = time.time()
step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
t = step.trainer
for c in range(cycles):
    if c % 25 == 24:                                   # a fresh trainer now and then: its modules' events are destroyed after their last graph
        del step, t
        gc.collect()
        step, _ = bench.build(dev, ("simclr", "byol", "barlow")[(c // 25) % 3], arch="resnet18", reduce_bottom_conv=True)
        t = step.trainer
    sg = StepGraph(t, mode="1", graph_floors=bool(c % 2) and "nofloors" not in VAR)
    for i in range(4):                                 # 2 eager steps, the capture + its replay, one more replay
        loss = sg(batches[i % 3])["loss"]
        t._after_step(i)
    if "keep" in VAR:
        KEEP.append(sg)
    else:
        sg.close()                                     # the graph is destroyed ...
    del sg
    for i in range(0 if "noeager" in VAR else 2):                                 # ... and the step runs eagerly on the same modules
        loss = t.train_step(batches[i])["loss"]
        t._after_step(i)
    if c % 10 == 0:
        print(c, round(loss, 4), round(time.time() - t0, 1), flush=True)
print("done", cycles, round(time.time() - t0, 1), flush=True)
