import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from conftest import seeded_randn
import test_gpu_graph as T
from ssv_amd.graph import StepGraph
dev=torch.device('cuda:0')
batches=T._batches(dev,6)
probe = seeded_randn(999, 48, 3, 32, 32).to(dev)
def run(mode, nsteps, with_probe, lrchange, ragged):
    t=T._trainer(dev,'simclr'); sg=StepGraph(t, mode='1' if mode=='graph' else '0', graph_floors=False)
    for i in range(nsteps):
        if lrchange and i and i % 20 == 0:
            for g in t.optim.param_groups: g['lr']*=0.7
        if with_probe and i % 10 == 5:
            with torch.no_grad(): t._features(probe)
        if ragged and i == 33:
            sg({k: v[:24] for k, v in batches[0].items()})
        sg(batches[i%6])
    torch.cuda.synchronize()
    return torch.cat([b.flatten().float() for n,b in t.encoder.named_buffers() if 'running' in n]), t.optim.arena.data.clone()
for cfg in ((60,True,False,False),(60,True,True,False),(60,True,False,True),(60,True,True,True)):
    e=run('eager',*cfg); e2=run('eager',*cfg); g=run('graph',*cfg); g2=run('graph',*cfg)
    print(cfg, 'eager==eager', torch.equal(e[0],e2[0]), 'graph==graph', torch.equal(g[0],g2[0]), 'eager==graph', torch.equal(e[0],g[0]), float((e[0]-g[0]).abs().max()))
