"""r05 diagnostic: per-tensor gradient error of the rank-k-of-8 step (tests/test_gpu_config3.py) and of dz itself."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import oracle
from conftest import seeded_randn
import test_gpu_config3 as T
from ssv_amd import distributed as hdist, nn as hnn, ops
from ssv_amd.utils import losses

dev = torch.device("cuda", 0)
WORLD, k = 8, 0
b, nglob, ld = 16, 128, 128
a1, a2 = seeded_randn(11, nglob, 3, 32, 32), seeded_randn(12, nglob, 3, 32, 32)
sh = lambda t, r: t[r * b:(r + 1) * b].to(dev)
enc, head, opt = T._build(dev)
blocks = torch.empty((WORLD, 2 * b, ld), device=dev)
with torch.no_grad():
    for r in range(WORLD):
        ops.l2norm_fwd(head(enc(sh(a1, r))).contiguous(), True, ld, out=blocks[r, :b])
        ops.l2norm_fwd(head(enc(sh(a2, r))).contiguous(), True, ld, out=blocks[r, b:])
zall = blocks.view(WORLD, 2, b, ld).permute(1, 0, 2, 3).reshape(2 * nglob, ld).contiguous()
packs = torch.empty((WORLD, 2 * b + 4), device=dev)
for r in range(WORLD):
    lse, pos = ops.ntxent_fwd(zall, nglob, b, r * b, 2.0)
    packs[r, :2 * b] = lse
    packs[r, 2 * b:] = ops.ntxent_loss(lse, pos, 1.0 / (2 * nglob))

def gather(out, mine):
    if out.shape == (WORLD * 2 * b, ld):
        out.view(WORLD, 2 * b, ld).copy_(blocks)
        out.view(WORLD, 2 * b, ld)[k].copy_(mine)
    else:
        out.copy_(packs)
        out[k].copy_(mine[0])

mode = sys.argv[1] if len(sys.argv) > 1 else "emu"
if mode == "emu":
    prev = hdist.emulate_world(WORLD, k, gather=gather, reduce=lambda t: t)
    hdist.attach_grad_sync(opt, [enc, head])
with hnn.parallel_views(dev) as pv:
    with pv.view(0):
        z1 = head(enc(sh(a1, k)))
    with pv.view(1):
        z2 = head(enc(sh(a2, k)))
z1.retain_grad(); z2.retain_grad()
if mode == "emu":
    loss = losses.SimclrLoss(True, 0.5)(z1, z2)
else:   # "manual": the loss gradient handed in from the oracle: isolates the encoder backward from the loss path
    loss = None
opt.zero_grad()
m32, m64 = T._oracle_pair()
res = {}
for name, m, cast in (("f32", m32, lambda t: t), ("f64", m64, lambda t: t.double())):
    zs = {}
    def emb(x, r, tag):
        z = m.embed(cast(x[r * b:(r + 1) * b]))
        if r == k:
            z.retain_grad(); zs[tag] = z
            return z
        return z.detach()
    ref = oracle.ntxent_loss(torch.cat([emb(a1, r, 1) for r in range(WORLD)]), torch.cat([emb(a2, r, 2) for r in range(WORLD)]), True, 0.5)
    ref.backward()
    res[name] = (ref.item(), [p.grad for p in m.params], zs[1].grad, zs[2].grad, zs[1].detach(), zs[2].detach())
if mode == "emu":
    loss.backward()
else:
    torch.autograd.backward([z1, z2], [res["f32"][2].to(dev), res["f32"][3].to(dev)])
hnn.join_view_streams(dev)
if mode == "emu":
    opt.grad_sync.finish()
    flat = opt.arena.grad.cpu()
else:
    flat = (opt.arena.grad + opt.arena.grad_alt).cpu()
torch.cuda.synchronize()
if mode == "emu":
    print("loss", loss.item(), res["f32"][0], res["f64"][0])
    rel = lambda a, c: float((a.double() - c.double()).norm() / c.double().norm())
    print("dz1 hip vs f64", rel(z1.grad.cpu(), res["f64"][2]), " f32 vs f64", rel(res["f32"][2], res["f64"][2]))
    print("dz2 hip vs f64", rel(z2.grad.cpu(), res["f64"][3]), " f32 vs f64", rel(res["f32"][3], res["f64"][3]))
    print("z1 hip vs f64", rel(z1.detach().cpu(), res["f64"][4]), " f32 vs f64", rel(res["f32"][4], res["f64"][4]))
e_hip, e_cpu = T._arena_vs_params(flat, res["f64"][1], res["f32"][1])
names = [n for n, _ in list(enc.named_parameters()) + list(head.named_parameters())]
for i, (eh, ec) in enumerate(zip(e_hip, e_cpu)):
    print(f"{i:3d} hip {eh:.2e} cpu {ec:.2e}")
