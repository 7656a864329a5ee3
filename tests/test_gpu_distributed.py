"""GPU: the data-parallel path with REAL kernels - two ranks sharing the one GPU of the test box (gloo transport, since
RCCL refuses two ranks on one device).  Checked against the oracle's emulation of the same semantics: every rank runs
its own shard with LOCAL BatchNorm statistics, NT-Xent sees the global batch, gradients are summed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle
from conftest import seeded_randn

pytestmark = pytest.mark.gpu
B, WORLD = 16, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                          LOCAL_RANK=str(rank), SSV_DIST_BACKEND="gloo")
        import torch.distributed as dist
        from ssv_amd import distributed as hdist, nn as hnn
        from ssv_amd.models import heads
        from ssv_amd.networks import resnet
        from ssv_amd.utils import losses, train_utils
        hdist.init_from_env()
        dev = torch.device("cuda", torch.cuda.current_device())
        torch.manual_seed(420)
        enc = resnet.resnet18(reduce_bottom_conv=True).to(dev)
        head = heads.SimclrProjectionHead(512, 128).to(dev)
        opt = train_utils.get_optimizer({"name": "sgd", "lr": 0.2, "weight_decay": 1e-4}, list(enc.parameters()) + list(head.parameters()))
        hdist.attach_grad_sync(opt)
        loss_fn = losses.SimclrLoss(True, 0.5)
        a1 = seeded_randn(1, B * WORLD, 3, 32, 32)[rank * B:(rank + 1) * B].to(dev)
        a2 = seeded_randn(2, B * WORLD, 3, 32, 32)[rank * B:(rank + 1) * B].to(dev)
        with hnn.parallel_views(dev) as pv:
            with pv.view(0):
                z1 = head(enc(a1))
            with pv.view(1):
                z2 = head(enc(a2))
        loss = loss_fn(z1, z2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        extra = _sharded_loss_checks(rank, dev, losses)
        q.put((rank, "ok" if extra is None else extra, loss.item(), opt.arena.data.cpu().numpy()))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def _sharded_loss_checks(rank, dev, losses):
    """BYOL pair MSE and Barlow Twins over a batch split across the two ranks == the oracle on the whole batch."""
    n, b = 2 * 24, 24
    sl = slice(rank * b, (rank + 1) * b)
    o = [seeded_randn(30 + k, n, 128).requires_grad_() for k in range(2)]
    t = [seeded_randn(32 + k, n, 128) for k in range(2)]
    ref = oracle.byol_mse_loss(o[0], o[1], t[0], t[1])
    ref.backward()
    ol = [x.detach()[sl].to(dev).requires_grad_() for x in o]
    loss = losses.byol_pair_loss(ol[0], ol[1], t[0][sl].to(dev), t[1][sl].to(dev))
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
    for k in range(2):
        np.testing.assert_allclose(ol[k].grad.cpu().numpy(), o[k].grad[sl].numpy(), rtol=1e-4, atol=1e-8)
    zi, zj = seeded_randn(40, n, 64), seeded_randn(41, n, 64)
    for normalize in (True, False):
        a, c = zi.clone().requires_grad_(), zj.clone().requires_grad_()
        ref = oracle.barlow_loss(a, c, normalize, 0.005)
        ref.backward()
        li, lj = zi[sl].to(dev).requires_grad_(), zj[sl].to(dev).requires_grad_()
        loss = losses.BarlowLoss(normalize, 0.005)(li, lj)
        loss.backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-4)
        scale = float(a.grad.abs().max())
        np.testing.assert_allclose(li.grad.cpu().numpy(), a.grad[sl].numpy(), rtol=2e-3, atol=2e-4 * scale)
        np.testing.assert_allclose(lj.grad.cpu().numpy(), c.grad[sl].numpy(), rtol=2e-3, atol=2e-4 * scale)
    return None


def test_two_ranks_match_oracle_data_parallel_emulation():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(WORLD)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for rank, msg, _, _ in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"
    (_, _, l0, p0), (_, _, l1, p1) = res
    assert l0 == l1, "every rank must return the same global-batch loss"
    np.testing.assert_array_equal(p0, p1)                      # identical replicas after the update
    # oracle: shards one after the other through one set of weights (BN per shard), loss on the concatenation
    m = oracle.SimCLROracle("resnet18", True, 128, lr=0.2, weight_decay=1e-4)
    a1, a2 = seeded_randn(1, B * WORLD, 3, 32, 32), seeded_randn(2, B * WORLD, 3, 32, 32)
    z1 = torch.cat([m.embed(a1[r * B:(r + 1) * B]) for r in range(WORLD)])
    z2 = torch.cat([m.embed(a2[r * B:(r + 1) * B]) for r in range(WORLD)])
    ref = oracle.ntxent_loss(z1, z2, True, 0.5)
    ref.backward()
    np.testing.assert_allclose(l0, ref.item(), rtol=1e-5)
    m._apply_sgd()
    # parameters after the step: compare tensor by tensor through the arena layout (ReLU-flip tolerant, see test_gpu_step)
    from ssv_amd.utils.train_utils import _ALIGN
    off, errs = 0, []
    for p in m.params:
        n = p.numel()
        got = torch.from_numpy(p0[off:off + n])
        got = got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)
        denom = float(p.detach().norm()) + 1e-12
        errs.append(float((got - p.detach()).norm()) / denom)
        off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
    # weights move by ~1e-3 of their norm in one step, BN biases start at 0 and ARE the (ReLU-flip-noisy) gradient
    assert np.median(errs) < 1e-4 and max(errs) < 2e-2, f"median {np.median(errs):.2e}, worst {max(errs):.2e}"
