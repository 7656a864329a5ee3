"""r05: is the host-heap corruption after destroying a two-stream HIP graph ours?  The same life cycle with PLAIN PyTorch kernels only (no libssv_hip.so): capture a step that
forks onto two side streams (event record / wait, as torch's own wait_stream does), replay it, destroy the graph, run the same work eagerly; many times.
    python tools/exp/r05_graph_torch_only_stress.py [cycles = 300] [streams = 2]"""
import faulthandler, gc, sys, time
faulthandler.enable()
import torch
dev = torch.device("cuda:0")
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
side = [torch.cuda.Stream() for _ in range(2)]
w = [torch.randn(512, 512, device=dev, requires_grad=True) for _ in range(6)]
x = torch.randn(256, 512, device=dev)


def step():
    main = torch.cuda.current_stream()
    outs = []
    for v in range(2):
        if nstreams > 1:
            side[v].wait_stream(main)
        with torch.cuda.stream(side[v] if nstreams > 1 else main):
            h = x
            for k in range(3):
                h = torch.relu(h @ w[3 * v + k])
            outs.append(h)
    if nstreams > 1:
        for v in range(2):
            main.wait_stream(side[v])
    loss = (outs[0] * outs[1]).mean()
    loss.backward()                                   # autograd runs each node on its forward's stream and joins the streams itself
    with torch.no_grad():
        for p in w:
            p -= 1e-3 * p.grad
            p.grad = None
    return loss


t0 = time.time()
for c in range(cycles):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    del g, loss                                       # the graph is destroyed ...
    gc.collect()
    for _ in range(2):                                # ... and the same work runs eagerly
        step()
    if c % 25 == 0:
        print(c, round(time.time() - t0, 1), flush=True)
torch.cuda.synchronize()
print("done", cycles, round(time.time() - t0, 1), flush=True)
