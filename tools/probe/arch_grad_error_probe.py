import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import oracle
from conftest import seeded_randn
from ssv_amd import ops
from ssv_amd.networks import resnet
dev = torch.device('cuda:0')
def run(arch, factory, w44, b=8, size=32):
    ops.WINOGRAD44 = w44
    torch.manual_seed(420)
    net = getattr(resnet, factory)(reduce_bottom_conv=True).to(dev)
    x, dy = seeded_randn(1700, b, 3, size, size), seeded_randn(1701, b, 2048)
    y = net(x.to(dev)); y.backward(dy.to(dev))
    def cpu(dtype):
        torch.manual_seed(420)
        p = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in oracle.init_resnet(arch, True).items()}
        for k, v in p.items():
            if v.dtype.is_floating_point and "running" not in k: v.requires_grad_(True)
        out = oracle.resnet_forward(p, x.to(dtype), arch, True); out.backward(dy.to(dtype)); return p, out.detach()
    p64, y64 = cpu(torch.float64); p32, y32 = cpu(torch.float32)
    e_cpu = float((y32.double()-y64).abs().max()); e_hip = float((y.detach().cpu().double()-y64).abs().max())
    errs = {n: float((p.grad.cpu().double()-p64[n].grad).norm()/(p64[n].grad.norm()+1e-30)) for n, p in net.named_parameters()}
    cpu_err = np.array([float((p32[k].grad.double()-p64[k].grad).norm()/(p64[k].grad.norm()+1e-30)) for k in errs])
    v = np.array(list(errs.values()))
    print(f"{arch} b={b} F44={w44}: features |err| hip {e_hip:.2e} cpu {e_cpu:.2e} | grad err median hip {np.median(v):.3e} cpu {np.median(cpu_err):.3e} ratio {np.median(v)/np.median(cpu_err):.2f} | max hip {v.max():.3e} cpu {cpu_err.max():.3e}", flush=True)
for arch, fac in (("wide_resnet50","wide_resnet50_2"),("resnet50","resnet50")):
    for b in (8, 32):
        for w44 in (False, True):
            run(arch, fac, w44, b=b)
