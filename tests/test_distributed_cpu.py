"""World-size-2 gloo tests (CPU) of the data-parallel host logic: row placement / all-gather of the
embeddings, the LSE exchange, loss all-reduce and the SUM gradient convention for all three losses (NT-Xent, BYOL pair MSE,
Barlow Twins).  The HIP kernels are
replaced, in these tests only, by oracle-backed CPU emulations of the same tensor-level entry points
(ssv_amd.ops.*), so what is exercised is the product's orchestration code in utils/losses.py and distributed.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import seeded_randn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ---- CPU emulations of the kernel entry points (restating include/ssv_hip.h semantics with the oracle) ------------
def _emu_l2norm_fwd(z, normalize=True, ldo=None, eps=1e-12, out=None):
    rows, d = z.shape
    ldo = d if ldo is None else ldo
    zhat = out if out is not None else torch.zeros(rows, ldo)
    norm = z.norm(dim=1, keepdim=True).clamp_min(eps) if normalize else torch.ones(rows, 1)
    zhat.zero_()
    zhat[:, :d] = z / norm
    return zhat, (1.0 / norm).flatten()


def _emu_l2norm_bwd(zhat, inv, dzhat, d, normalize=True):
    zh, dh = zhat[:, :d], dzhat[:, :d]
    if not normalize:
        return dh.clone()
    return (dh - zh * (zh * dh).sum(1, keepdim=True)) * inv[:, None]


def _rows(nglob, b, seg0):
    return torch.cat([torch.arange(seg0, seg0 + b), nglob + torch.arange(seg0, seg0 + b)])


def _emu_ntxent_fwd(zall, nglob, b, seg0, inv_temp):
    r = _rows(nglob, b, seg0)
    s = (zall[r] @ zall.t()) * inv_temp
    pos = s[torch.arange(2 * b), (r + nglob) % (2 * nglob)]
    s[torch.arange(2 * b), r] = float("-inf")
    return torch.logsumexp(s, 1), pos


def _emu_ntxent_loss(lse, pos, scale):
    return ((lse - pos).sum() * scale).reshape(())


def _emu_ntxent_bwd(zall, lse_all, nglob, b, seg0, inv_temp, gscale):
    r = _rows(nglob, b, seg0)
    s = (zall[r] @ zall.t()) * inv_temp
    w = torch.exp(s - lse_all[r][:, None]) + torch.exp(s - lse_all[None, :])
    w[torch.arange(2 * b), r] = 0.0
    return gscale * (w @ zall - 2.0 * zall[(r + nglob) % (2 * nglob)])


def _emu_mse_pair(o1, o2, t1, t2, scale):
    loss = scale * (((o1 - t2) ** 2).sum() + ((o2 - t1) ** 2).sum())
    return loss.reshape(()), 2 * scale * (o1 - t2), 2 * scale * (o2 - t1)


def _emu_bn_train_fwd(x, gamma, beta, running_mean, running_var, nbt, relu=False, residual=None, eps=1e-5, **_):
    mean, var = x.mean(0), x.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + eps)
    return (x - mean) * invstd * gamma + beta, mean, invstd


def _emu_bn_train_bwd(dy, y, x, gamma, mean, invstd, relu, dgamma, dbeta, want_dres=False, accumulate=True, relu_mask=None):
    xh = (x - mean) * invstd
    dgamma.copy_((dy * xh).sum(0))
    dbeta.copy_(dy.sum(0))
    m = x.shape[0]
    return gamma * invstd * (dy - dbeta / m - xh * dgamma / m), None


def _emu_conv1x1_wgrad(x, dy, w_like, dw, stride=1, pad=0, accumulate=True):
    k, c = dy.shape[-1], x.shape[-1]
    dw.copy_(dy.reshape(-1, k).t() @ x.reshape(-1, c))              # dw[k][c] = sum_m dy[m,k] x[m,c]
    return dw


def _emu_conv1x1_fwd(x, w, stride=1, pad=0, bias=None, addend=None):
    return (x.reshape(-1, x.shape[-1]) @ w.t()).reshape(*x.shape[:-1], w.shape[0])


def _emu_conv1x1_dgrad(dy, w, x_shape, stride=1, pad=0, addend=None, out=None):
    return (dy.reshape(-1, dy.shape[-1]) @ w).reshape(tuple(x_shape))


def _emu_barlow_cgrad(craw, inv_b, lmbda):
    c = craw * inv_b
    eye = torch.eye(c.shape[0])
    wgt = eye + lmbda * (1 - eye)
    return (wgt * (c - eye) ** 2).sum().reshape(()), 2 * wgt * (c - eye) * inv_b


def _emu_dino_loss(teacher, student, center, temp_s, temp_t, weight, loss, accumulate):
    from oracle import vit as ovit
    with torch.enable_grad():                                    # called from inside autograd.Function.forward
        st = student.clone().requires_grad_()
        val = weight * ovit.dino_loss(teacher, st, temp_s, temp_t, center.view(1, -1))
        val.backward()
    loss.copy_(loss + val.detach() if accumulate else val.detach())
    return st.grad


def _patch_ops():
    from ssv_amd import ops
    ops.l2norm_fwd, ops.l2norm_bwd = _emu_l2norm_fwd, _emu_l2norm_bwd
    ops.ntxent_fwd, ops.ntxent_loss, ops.ntxent_bwd = _emu_ntxent_fwd, _emu_ntxent_loss, _emu_ntxent_bwd
    ops.scale_ = lambda x, f: x.mul_(f)
    ops.fill_ = lambda x, v: x.fill_(v)
    ops.add_ = lambda dst, src: dst.add_(src)
    ops.mse_pair, ops.barlow_cgrad = _emu_mse_pair, _emu_barlow_cgrad
    ops.dino_loss = _emu_dino_loss
    ops.bn_train_fwd, ops.bn_train_bwd = _emu_bn_train_fwd, _emu_bn_train_bwd
    ops.conv2d_wgrad, ops.conv2d_fwd, ops.conv2d_dgrad = _emu_conv1x1_wgrad, _emu_conv1x1_fwd, _emu_conv1x1_dgrad


class _StubLoader:
    """GpuTwoViewLoader with the HIP view kernels stubbed out: what runs is the product's permutation / sharding / step logic."""

    def __new__(cls, n, batch, **kw):
        from ssv_amd.utils import data_utils

        class L(data_utils.GpuTwoViewLoader):
            def _setup_transforms(self, transforms):
                pass

            def _make(self, idx, step):
                return {"index": idx, "step": step}
        return L(np.zeros((n, 2, 2, 3), np.uint8), np.arange(n) % 5, None, batch, True, torch.device("cpu"), **kw)


def _loader_epochs(loader, epochs):
    out = []
    for _ in range(epochs):
        first = loader.step
        batches = list(loader)
        assert [b["step"] for b in batches] == list(range(first, first + len(batches))) and len(batches) == len(loader)
        out.append([b["index"].tolist() for b in batches])
    return out


class _FakeArena:
    """ParamArena's layout (64-float aligned runs of one flat buffer, two gradient slabs) on the CPU."""

    def __init__(self, params, rank, salt):
        self.params, self.offsets, off = list(params), [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        g = torch.Generator().manual_seed(1000 * salt + rank)
        self._grads = torch.randn(2 * off, generator=g)
        self.grad, self.grad_alt = self._grads[:off], self._grads[off:]


class _StagedNet(torch.nn.Module):
    """Three "stages" of parameters + a head that belongs to no stage, like ResNet + its bridged wrapper."""

    def __init__(self):
        super().__init__()
        self.s0 = torch.nn.Linear(7, 5)
        self.s1 = torch.nn.Linear(5, 33)
        self.s2 = torch.nn.Linear(33, 9)
        self.head = torch.nn.Linear(9, 3)

    def grad_stages(self):
        return [list(self.s0.parameters()), list(self.s1.parameters()), list(self.s2.parameters())]


def _bucketed_exchange_checks(rank, world, hdist):
    from types import SimpleNamespace
    net, extra = _StagedNet(), torch.nn.Linear(4, 4)                      # `extra`: owned by the optimizer, known to no module
    params = list(net.parameters()) + list(extra.parameters())
    single = SimpleNamespace(arena=_FakeArena(params, rank, 1))
    bucketed = SimpleNamespace(arena=_FakeArena(params, rank, 1))
    assert torch.equal(single.arena._grads, bucketed.arena._grads)
    want = single.arena.grad + single.arena.grad_alt                      # this rank's fold; summed over ranks below
    gathered = [torch.zeros_like(want) for _ in range(world)]
    dist.all_gather(gathered, want)
    want = gathered[0] + gathered[1]
    hdist.BucketedGradSync(single, (), bucketed=False).finish()
    sync = hdist.BucketedGradSync(bucketed, [net], bucketed=True)
    assert [b[0] for b in sync.buckets] == ["_StagedNet.stage0", "_StagedNet.stage1", "_StagedNet.stage2", "_StagedNet.rest"]
    assert [(lo, hi) for _, lo, hi in sync.buckets] == [(0, 128), (128, 384), (384, 768), (768, 896)] and bucketed.arena.numel == 1024
    sync.expect(net)
    sync.expect(net)                                                      # two backward passes (the two views) will report
    order = []
    for _ in range(2):
        for stage in (2, 1, 0, "rest"):                                   # backward order: the last stage completes first
            assert sync.has(net, stage) and not sync.has(extra, 0)
            before = len(sync.launched)
            sync.ready(net, stage)
            if len(sync.launched) > before:
                order.append(stage)
    assert order == [2, 1, 0, "rest"]                                     # each bucket goes out when the SECOND pass reports it
    sync.ready(net, 2)                                                    # a late / duplicate report is ignored
    assert not torch.equal(bucketed.arena.grad[896:], single.arena.grad[896:])       # `extra` not exchanged yet ...
    sync.finish()                                                         # ... finish() picks it up
    assert torch.equal(bucketed.arena.grad, single.arena.grad), "bucketed exchange differs from the single call"
    np.testing.assert_allclose(single.arena.grad.numpy(), want.numpy(), rtol=0, atol=0)
    assert not sync.pending and not sync.launched
    # a step in which only one pass reports (the other pass's backward never ran): nothing may be lost or reduced twice
    third = SimpleNamespace(arena=_FakeArena(params, rank, 2))
    ref = SimpleNamespace(arena=_FakeArena(params, rank, 2))
    hdist.BucketedGradSync(ref, (), bucketed=False).finish()
    sync = hdist.BucketedGradSync(third, [net], bucketed=True)
    sync.expect(net)
    sync.expect(net)
    for stage in (2, 1, 0, "rest"):
        sync.ready(net, stage)
    assert not sync.launched
    sync.finish()
    assert torch.equal(third.arena.grad, ref.arena.grad)
    # a staged sub-module that is ALSO reachable through another outer module (aliasing): the pass through that other module never expect()ed the
    # sub-module's buckets, so its report must not count them down (it would launch the all-reduce before the owner's second view has accumulated)
    fourth = SimpleNamespace(arena=_FakeArena(params, rank, 3))
    sync = hdist.BucketedGradSync(fourth, [net], bucketed=True)
    mine = sync.expect(net)
    assert mine == frozenset(range(4))
    sync.expect(net)
    sync.ready(net, 2, expected=frozenset())                              # a pass that was not counted for bucket 2 reports it: ignored
    sync.ready(net, 2, expected=mine)
    assert not sync.launched                                              # still one counted pass outstanding
    sync.ready(net, 2, expected=mine)
    assert sync.launched == {2}
    sync.finish()
    # a run of parameters lying strictly INSIDE an existing bucket is not a second bucket (interval intersection, not end-point containment)
    class _Inner(torch.nn.Module):
        def __init__(self, outer):
            super().__init__()
            self.inner = outer.s1                                         # s1's span (128, 384) lies inside a bucket made of s0..s2

        def grad_stages(self):
            return [list(self.inner.parameters())]

    class _Wide(torch.nn.Module):
        def __init__(self, outer):
            super().__init__()
            self.outer = outer

        def grad_stages(self):
            return [list(self.outer.s0.parameters()) + list(self.outer.s1.parameters()) + list(self.outer.s2.parameters())]
    wide, inner = _Wide(net), _Inner(net)
    fifth = SimpleNamespace(arena=_FakeArena(params, rank, 4))
    sync = hdist.BucketedGradSync(fifth, [wide, inner], bucketed=True)
    spans = sorted((lo, hi) for _, lo, hi in sync.buckets)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans     # disjoint
    assert (0, 768) in spans and (128, 384) not in spans


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ssv_amd import distributed as hdist
    from ssv_amd.utils import losses
    hdist.init_from_env(backend="gloo")
    assert hdist.is_on() and hdist.world_size() == world and hdist.rank() == rank
    _patch_ops()
    try:
        # ---- 1. all_gather_rows: every rank's block lands in its slot
        buf = torch.zeros(world * 3, 2)
        buf[rank * 3:(rank + 1) * 3] = rank + 1
        hdist.all_gather_rows(buf, 3)
        assert torch.equal(buf[:, 0], torch.arange(1, world + 1).repeat_interleave(3).float())
        # ---- 2. sharded NT-Xent == global NT-Xent (loss identical on every rank; dz rows = rows of the global gradient)
        n, d = 12, 20                                             # d is padded to 32 inside
        b = n // world
        zi, zj = seeded_randn(1, n, d), seeded_randn(2, n, d)
        a, c = zi.clone().requires_grad_(), zj.clone().requires_grad_()
        ref = oracle.ntxent_loss(a, c, True, 0.5)
        ref.backward()
        li, lj = zi[rank * b:(rank + 1) * b].clone().requires_grad_(), zj[rank * b:(rank + 1) * b].clone().requires_grad_()
        loss = losses.SimclrLoss(True, 0.5)(li, lj)
        (loss * 3.0).backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
        np.testing.assert_allclose(li.grad.numpy() / 3.0, a.grad[rank * b:(rank + 1) * b].numpy(), rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(lj.grad.numpy() / 3.0, c.grad[rank * b:(rank + 1) * b].numpy(), rtol=2e-4, atol=1e-7)
        # ---- 3. data-parallel step semantics: local BN per rank and view, global loss, SUM of gradients.
        torch.manual_seed(0)
        w1, w2 = torch.randn(8, 6), torch.randn(6, 8)
        gam = torch.ones(8)

        def net(x, w1_, w2_, g_):                                 # Linear -> BN(train, local stats) -> ReLU -> Linear
            h = x @ w1_.t()
            h = (h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + 1e-5) * g_
            return torch.relu(h) @ w2_.t()

        x1, x2 = seeded_randn(3, n, 6), seeded_randn(4, n, 6)
        # single-process emulation: shards run one after the other through the same weights, loss on the concatenation
        ps = [t.clone().requires_grad_() for t in (w1, w2, gam)]
        z1 = torch.cat([net(x1[r * b:(r + 1) * b], *ps) for r in range(world)])
        z2 = torch.cat([net(x2[r * b:(r + 1) * b], *ps) for r in range(world)])
        oracle.ntxent_loss(z1, z2, True, 0.5).backward()
        # this rank
        pl = [t.clone().requires_grad_() for t in (w1, w2, gam)]
        l = losses.SimclrLoss(True, 0.5)(net(x1[rank * b:(rank + 1) * b], *pl), net(x2[rank * b:(rank + 1) * b], *pl))
        l.backward()
        flat = torch.cat([p.grad.flatten() for p in pl])
        hdist.all_reduce_sum(flat)                                # what FusedSGD.grad_sync does on the arena
        np.testing.assert_allclose(flat.numpy(), torch.cat([p.grad.flatten() for p in ps]).numpy(), rtol=5e-4, atol=1e-6)
        # ---- 4. BYOL pair loss: global mean on every rank, SUM of per-rank gradients = gradient of the global mean
        o = [seeded_randn(10 + k, n, d).requires_grad_() for k in range(2)]
        t = [seeded_randn(12 + k, n, d) for k in range(2)]
        ref = oracle.byol_mse_loss(o[0], o[1], t[0], t[1])
        ref.backward()
        sl = slice(rank * b, (rank + 1) * b)
        ol = [x.detach()[sl].clone().requires_grad_() for x in o]
        loss = losses.byol_pair_loss(ol[0], ol[1], t[0][sl].clone(), t[1][sl].clone())
        loss.backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
        for k in range(2):
            np.testing.assert_allclose(ol[k].grad.numpy(), o[k].grad[sl].numpy(), rtol=1e-5, atol=1e-8)
        # ---- 5. Barlow Twins: statistics and the cross-correlation are over the GLOBAL batch
        dd = 16
        zi, zj = seeded_randn(20, n, dd), seeded_randn(21, n, dd)
        for normalize in (True, False):
            a, c = zi.clone().requires_grad_(), zj.clone().requires_grad_()
            ref = oracle.barlow_loss(a, c, normalize, 0.005)
            ref.backward()
            li, lj = zi[sl].clone().requires_grad_(), zj[sl].clone().requires_grad_()
            loss = losses.BarlowLoss(normalize, 0.005)(li, lj)
            loss.backward()
            np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-5)
            np.testing.assert_allclose(li.grad.numpy(), a.grad[sl].numpy(), rtol=2e-3, atol=2e-6)
            np.testing.assert_allclose(lj.grad.numpy(), c.grad[sl].numpy(), rtol=2e-3, atol=2e-6)
        # ---- 6. DINO: every rank scores its own shard (the reference's cat-then-view grouping is per batch, so the data-parallel
        #         loss is DEFINED as the mean of the per-shard reference losses); SUM of gradients = gradient of that mean
        from oracle import vit as ovit
        from ssv_amd.models import dino as hdino
        bs, vg, vl, k = 3, 2, 4, 24
        center = seeded_randn(30, k)
        shards = [[seeded_randn(31 + 10 * r + i, n_, k) for i, n_ in enumerate((2 * bs * vg, 2 * bs * vl, 2 * bs * vg))] for r in range(world)]

        def shard_loss(sg, sl, tg):
            ng, nl = bs * vg, bs * vl
            s1 = torch.cat((sg[:ng], sl[:nl]), 0).view(bs, vg + vl, k)
            s2 = torch.cat((sg[ng:], sl[nl:]), 0).view(bs, vg + vl, k)
            t1, t2 = tg[:ng].view(bs, vg, k), tg[ng:].view(bs, vg, k)
            return 0.5 * ovit.dino_loss(t1, s2, 0.1, 0.04, center.view(1, -1)) + 0.5 * ovit.dino_loss(t2, s1, 0.1, 0.04, center.view(1, -1))

        refs = []
        for r in range(world):
            sg, sl = shards[r][0].clone().requires_grad_(), shards[r][1].clone().requires_grad_()
            (shard_loss(sg, sl, shards[r][2]) / world).backward()
            refs.append((sg.grad, sl.grad))
        want = sum(float(shard_loss(*shards[r])) for r in range(world)) / world
        sg, sl = shards[rank][0].clone().requires_grad_(), shards[rank][1].clone().requires_grad_()
        loss = hdino._DinoLossFn.apply(sg, sl, shards[rank][2], center, bs, vg, vl, 0.1, 0.04)
        loss.backward()
        np.testing.assert_allclose(loss.item(), want, rtol=2e-6)
        np.testing.assert_allclose(sg.grad.numpy(), refs[rank][0].numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(sl.grad.numpy(), refs[rank][1].numpy(), rtol=1e-5, atol=1e-9)
        # ---- 7. the CLI's loaders shard every global batch: the ranks' index sets are disjoint, their union is the global batch
        #         of the SHARED permutation, the step counter (augmentation stream key) is global; broadcast_object / barrier
        idx_sets = _loader_epochs(_StubLoader(50, 8), epochs=2)
        gathered = [None] * world
        dist.all_gather_object(gathered, idx_sets)
        perm_gen = torch.Generator().manual_seed(420)
        for epoch in range(2):
            order = torch.randperm(50, generator=perm_gen)
            steps = gathered[0][epoch]
            assert len(steps) == len(gathered[1][epoch]) == 4                      # 3 full global batches of 16 + a ragged one of 2
            for st in range(4):
                a, b_ = gathered[0][epoch][st], gathered[1][epoch][st]
                assert not set(a) & set(b_)
                assert a + b_ == order[16 * st:16 * st + len(a) + len(b_)].tolist()   # rank r owns rows [r*B, (r+1)*B) of the global batch
                assert len(a) == len(b_) == (8 if st < 3 else 1)
        assert hdist.broadcast_object(f"run-of-rank-{rank}") == "run-of-rank-0"
        # ---- 8. bucketed gradient exchange (collective 2): buckets reported stage by stage by two backward passes (two
        #         gradient slabs) == ONE call over the whole arena, bit for bit; parameters outside every bucket are picked up by finish()
        _bucketed_exchange_checks(rank, world, hdist)
        hdist.barrier()
        out.put((rank, "ok"))
    except Exception as e:                                        # surface the failure to the parent
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_loss_and_gradient_sum():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_single_process_defaults():
    from ssv_amd import distributed as hdist
    assert not hdist.is_on() and hdist.world_size() == 1 and hdist.rank() == 0
    t = torch.ones(3)
    assert hdist.all_reduce_sum(t) is t and hdist.all_gather_rows(t, 3) is t


def test_loader_sharding_is_a_partition_of_the_single_process_epoch():
    """World sizes 1, 2, 3, 8 without a process group (rank / world passed in): each global batch is split into equal disjoint
    shards in rank order; world 1 keeps the reference's ragged last batch; at most world-1 samples are left out per epoch."""
    n, b = 103, 8
    single = _loader_epochs(_StubLoader(n, b, rank=0, world=1), 1)[0]
    assert sum(len(x) for x in single) == n and len(single[-1]) == n % b                 # last batch NOT dropped (utils/data_utils.py:119)
    order = [i for batch in single for i in batch]
    assert sorted(order) == list(range(n))
    for world in (2, 3, 8):
        per_rank = [_loader_epochs(_StubLoader(n, b, rank=r, world=world), 1)[0] for r in range(world)]
        assert len({len(x) for x in per_rank}) == 1                                      # every rank runs the same number of steps
        seen = []
        for st in range(len(per_rank[0])):
            shard_sizes = {len(per_rank[r][st]) for r in range(world)}
            assert len(shard_sizes) == 1                                                 # equal shards: what the all-gather needs
            glob = [i for r in range(world) for i in per_rank[r][st]]
            assert glob == order[len(seen):len(seen) + len(glob)]                        # same permutation as the single-process epoch
            seen += glob
        assert len(set(seen)) == len(seen) and n - len(seen) < world
    assert _StubLoader(n, b, rank=0, world=1).num_classes == 5


def _patch_loss_ops(monkeypatch):
    from ssv_amd import ops
    for name, fn in (("l2norm_fwd", _emu_l2norm_fwd), ("l2norm_bwd", _emu_l2norm_bwd), ("ntxent_fwd", _emu_ntxent_fwd), ("ntxent_loss", _emu_ntxent_loss),
                     ("ntxent_bwd", _emu_ntxent_bwd), ("scale_", lambda x, f: x.mul_(f)), ("fill_", lambda x, v: x.fill_(v)), ("add_", lambda d, s_: d.add_(s_)),
                     ("mse_pair", _emu_mse_pair)):
        monkeypatch.setattr(ops, name, fn)


def test_emulated_world_replicated_peers_equal_the_oracle_on_the_repeated_batch(monkeypatch):
    """ssv_amd.distributed.emulate_world(W) - bench.py's config3_rank_emulation: ONE process takes the data-parallel code paths of rank r of W with the
    transport replaced; with the default peers (every rank holds this rank's shard) the loss and the local gradient rows are exactly those of the
    W x repeated batch, and the emulated SUM all-reduce multiplies by W."""
    from ssv_amd import distributed as hdist
    from ssv_amd.utils import losses
    _patch_loss_ops(monkeypatch)
    world, rank, b, d = 4, 2, 6, 20
    zi, zj = seeded_randn(1, b, d), seeded_randn(2, b, d)
    a, c = zi.repeat(world, 1).requires_grad_(), zj.repeat(world, 1).requires_grad_()
    ref = oracle.ntxent_loss(a, c, True, 0.5)
    ref.backward()
    assert not hdist.is_on()
    prev = hdist.emulate_world(world, rank)
    try:
        assert hdist.is_on() and hdist.emulated() and hdist.world_size() == world and hdist.rank() == rank
        assert hdist.init_from_env() == (rank, world)                      # no process group is started under an emulation
        li, lj = zi.clone().requires_grad_(), zj.clone().requires_grad_()
        loss = losses.SimclrLoss(True, 0.5)(li, lj)
        loss.backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
        sl = slice(rank * b, (rank + 1) * b)
        np.testing.assert_allclose(li.grad.numpy(), a.grad[sl].numpy(), rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(lj.grad.numpy(), c.grad[sl].numpy(), rtol=2e-4, atol=1e-7)
        t = torch.arange(5.0)
        assert torch.equal(hdist.all_reduce_sum(t), torch.arange(5.0) * world)
        buf = torch.zeros(world * 3, 2)
        buf[rank * 3:(rank + 1) * 3] = 7.0
        assert torch.equal(hdist.all_gather_rows(buf, 3), torch.full((world * 3, 2), 7.0))
        hdist.barrier()                                                    # group-only helpers are no-ops under an emulation
        assert hdist.broadcast_object("x") == "x"
        # the BYOL pair loss: global mean with the scalar summed over the (identical) ranks
        o = [seeded_randn(10 + k, b, d) for k in range(2)]
        tg = [seeded_randn(12 + k, b, d) for k in range(2)]
        want = oracle.byol_mse_loss(o[0], o[1], tg[0], tg[1])
        got = losses.byol_pair_loss(o[0].clone().requires_grad_(), o[1].clone().requires_grad_(), tg[0], tg[1])
        np.testing.assert_allclose(got.item(), want.item(), rtol=2e-6)
    finally:
        hdist.restore_world(prev)
    assert not hdist.is_on() and hdist.world_size() == 1


def test_emulated_world_with_real_peers_is_the_sharded_loss(monkeypatch):
    """The ``gather`` closure supplies REAL peers (what tests/test_gpu_config3.py does with shards run one after the other): rank r's loss and gradient
    rows equal the oracle's on the concatenated batch."""
    from ssv_amd import distributed as hdist
    from ssv_amd.utils import losses
    _patch_loss_ops(monkeypatch)
    world, rank, b, d, ld = 4, 1, 5, 20, 32
    n = world * b
    zi, zj = seeded_randn(3, n, d), seeded_randn(4, n, d)
    a, c = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    ref = oracle.ntxent_loss(a, c, True, 0.5)
    ref.backward()
    blocks = torch.zeros(world, 2 * b, ld)
    for r in range(world):
        _emu_l2norm_fwd(zi[r * b:(r + 1) * b], True, ld, out=blocks[r, :b])
        _emu_l2norm_fwd(zj[r * b:(r + 1) * b], True, ld, out=blocks[r, b:])
    zall = blocks.view(world, 2, b, ld).permute(1, 0, 2, 3).reshape(2 * n, ld)
    packs = torch.zeros(world, 2 * b + 4)
    for r in range(world):
        lse, pos = _emu_ntxent_fwd(zall, n, b, r * b, 2.0)
        packs[r, :2 * b], packs[r, 2 * b:] = lse, _emu_ntxent_loss(lse, pos, 1.0 / (2 * n))

    def gather(out, mine):
        src = blocks.view(world * 2 * b, ld) if out.shape[-1] == ld else packs
        out.copy_(src)
        out.view(world, *mine.shape)[rank].copy_(mine)
    prev = hdist.emulate_world(world, rank, gather=gather)
    try:
        sl = slice(rank * b, (rank + 1) * b)
        li, lj = zi[sl].clone().requires_grad_(), zj[sl].clone().requires_grad_()
        loss = losses.SimclrLoss(True, 0.5)(li, lj)
        loss.backward()
    finally:
        hdist.restore_world(prev)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
    np.testing.assert_allclose(li.grad.numpy(), a.grad[sl].numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(lj.grad.numpy(), c.grad[sl].numpy(), rtol=2e-4, atol=1e-7)
    with pytest.raises(ValueError):
        hdist.emulate_world(4, 4)
