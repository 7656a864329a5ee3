"""Diagnostic: per-tensor gradient error of the HIP step vs the fp32 oracle and vs an fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle
from conftest import seeded_randn
from test_gpu_step import _Step, rel_l2

arch, rbc, B, S = (sys.argv[1], sys.argv[2] == "1", int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("resnet18", True, 64, 32)
dev = torch.device("cuda:0")
SEED = int(os.environ.get("SEED", "100"))
a1, a2 = seeded_randn(SEED, B, 3, S, S), seeded_randn(SEED + 1, B, 3, S, S)
m = _Step(dev, arch, rbc)
loss, z1, z2 = m.step(a1, a2)
o32 = oracle.SimCLROracle(arch, rbc, 128, lr=0.2, weight_decay=1e-4)
r32 = o32.train_step(a1, a2, return_z=True)
torch.set_default_dtype(torch.float64)
o64 = oracle.SimCLROracle(arch, rbc, 128, lr=0.2, weight_decay=1e-4)
# same weights as fp32 init, promoted
for d64, d32 in ((o64.encoder, None), (o64.proj_head, None)):
    pass
torch.set_default_dtype(torch.float32)
o32b = oracle.SimCLROracle(arch, rbc, 128, lr=0.2, weight_decay=1e-4)
for dst, src in ((o64.encoder, o32b.encoder), (o64.proj_head, o32b.proj_head)):
    for k in dst:
        if dst[k].dtype.is_floating_point:
            dst[k].data = src[k].detach().double()
r64 = o64.train_step(a1.double(), a2.double(), return_z=True)
print("loss hip %.9f cpu32 %.9f cpu64 %.9f" % (loss, r32["loss"], r64["loss"]))
print("z err hip-vs-64 %.3e  cpu32-vs-64 %.3e" % (float((z1.cpu().double() - r64["z_1"]).abs().max()), float((r32["z_1"].double() - r64["z_1"]).abs().max())))
keys = [k for k in list(o32.encoder.keys()) + ["head." + k for k in o32.proj_head.keys()] if k.endswith(".weight") or k.endswith(".bias")]
print("%-34s %-18s %10s %10s %10s" % ("tensor", "shape", "hip/64", "cpu32/64", "hip/cpu32"))
for k, p, g32, g64, off in zip(keys, m.params(), o32.last_grads, o64.last_grads, m.optim.arena.offsets):
    got = m.grads[off:off + p.numel()]
    got = got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)
    e1, e2, e3 = rel_l2(got, g64), rel_l2(g32, g64), rel_l2(got, g32)
    flag = " <<<" if e1 > 5 * e2 + 1e-6 else ""
    print("%-34s %-18s %10.2e %10.2e %10.2e%s" % (k, tuple(p.shape), e1, e2, e3, flag))
