"""GPU: BASELINE config 3 ("SimCLR resnet50 global batch 4096 on 8xMI355X") at the shape ONE RANK of it executes, on the one device a
test box has: local batch 512 per view, Nglob = 4096 - 1,024 local rows of NT-Xent against 8,192 gathered columns.

  * the eight row blocks through the C ABI (unsplit and column-split kernels) against the oracle's loss on the FULL batch
    (utils/losses.py:15-46 restated in oracle/losses.py): partial losses sum to the loss, dZ blocks tile the global dZ;
  * one resnet18 step of "rank k of 8" under ssv_amd.distributed.emulate_world with REAL peers - the other seven shards run one after
    the other through the same weights (BatchNorm statistics per shard and view, as every rank would compute them), their normalised
    embeddings / row log-sum-exps / loss partials are what the emulated all-gathers deliver - against the oracle's data-parallel
    emulation (models/simclr.py:86-95 applied to the concatenated batch with per-shard BatchNorm): the loss, and rank k's own share of
    the gradient (what it contributes to the SUM all-reduce);
  * the replicated-peer emulation bench.py --emulate-world uses (all ranks hold the same shard) equals the oracle on the 8x repeated batch.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import seeded_randn
from test_gpu_ops import close

pytestmark = pytest.mark.gpu
WORLD = 8


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def test_default_splits_fill_the_device_at_the_config3_shape():
    from ssv_amd import _lib
    lib = _lib.load()
    s = lib.ssv_ntxent_default_splits(4096, 512)
    assert s * (2 * 512 // 32) >= 256, s                      # >= one workgroup per CU of an MI355X
    assert (2 * 4096 // 32) // s >= 8                          # every split still sweeps >= 8 column tiles
    assert lib.ssv_ntxent_default_splits(64, 64) == 1          # config 1: 4 column tiles - nothing to split
    assert lib.ssv_ntxent_split_workspace_bytes(512, 128, s) == s * 1024 * 128 * 4
    assert lib.ssv_ntxent_split_workspace_bytes(512, 128, 1) == 0


@pytest.mark.parametrize("nglob,b,seg0,d", [(96, 24, 48, 128), (50, 50, 0, 64), (200, 40, 120, 128), (33, 11, 22, 96)])
@pytest.mark.parametrize("splits", [2, 3, 7, 64])
def test_ntxent_column_splits_on_ragged_shapes(dev, nglob, b, seg0, d, splits):
    """The column-split row kernels on shapes that are no multiple of anything - row blocks of 48 / 100 / 80 / 22 rows (partial 32-row tiles), 2 Nglob = 192 / 100 / 400 / 66
    columns (partial last column tile, a split count that leaves trailing splits empty: dropped; more splits than column tiles: refused), a rank's block in the middle of the batch - against the unsplit
    kernels (the same sums regrouped: rounding level) and, through them, the oracle on the full batch."""
    from ssv_amd import _lib, ops
    zi, zj = seeded_randn(300 + nglob, nglob, d), seeded_randn(301 + nglob, nglob, d)
    zh_i, zh_j = oracle.l2_normalize(zi).detach().requires_grad_(), oracle.l2_normalize(zj).detach().requires_grad_()
    oracle.ntxent_loss(zh_i, zh_j, False, 0.5).backward()
    zall = torch.cat([zh_i.detach(), zh_j.detach()]).to(dev).contiguous()
    inv_t, gscale = 2.0, 2.0 / (2 * nglob)
    lse1, pos1 = ops.ntxent_fwd(zall, nglob, b, seg0, inv_t, splits=1)
    if splits > -(-2 * nglob // 32):                           # more splits than 32-column tiles: refused, not silently clamped
        with pytest.raises(_lib.SsvError, match="more splits"):
            ops.ntxent_fwd(zall, nglob, b, seg0, inv_t, splits=splits)
        return
    lse_s, pos_s = ops.ntxent_fwd(zall, nglob, b, seg0, inv_t, splits=splits)
    close(lse_s, lse1, rtol=2e-6, what="lse split vs unsplit")
    close(pos_s, pos1, rtol=2e-6, what="positive logit split vs unsplit")
    # every row's log-sum-exp (the backward needs all of them): one call per block of b rows, unsplit
    lse_all = torch.empty(2 * nglob, device=dev)
    for r0 in range(0, nglob, b):
        bb = min(b, nglob - r0)
        l, _ = ops.ntxent_fwd(zall, nglob, bb, r0, inv_t, splits=1)
        lse_all[r0:r0 + bb], lse_all[nglob + r0:nglob + r0 + bb] = l[:bb], l[bb:]
    dz1 = ops.ntxent_bwd(zall, lse_all, nglob, b, seg0, inv_t, gscale, splits=1)
    dzs = ops.ntxent_bwd(zall, lse_all, nglob, b, seg0, inv_t, gscale, splits=splits)
    close(dzs, dz1, rtol=1e-4, what="dZ split vs unsplit")
    close(dzs[:b], zh_i.grad[seg0:seg0 + b], rtol=5e-4, what="dzi vs oracle")
    close(dzs[b:], zh_j.grad[seg0:seg0 + b], rtol=5e-4, what="dzj vs oracle")


def test_ntxent_eight_rank_blocks_at_config3_shape_match_the_full_batch_oracle(dev):
    """world 8, Nglob 4096, Bloc 512, D 128: ssv_ntxent_fwd[_split] / _loss / _bwd[_split] per row block vs oracle.ntxent_loss on the full batch."""
    from ssv_amd import _lib, ops
    nglob, d, b = 4096, 128, 512
    zi, zj = seeded_randn(191, nglob, d), seeded_randn(192, nglob, d)
    a, c = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    ref = oracle.ntxent_loss(a, c, True, 0.5)
    zh_i, zh_j = oracle.l2_normalize(zi).detach().requires_grad_(), oracle.l2_normalize(zj).detach().requires_grad_()
    oracle.ntxent_loss(zh_i, zh_j, False, 0.5).backward()      # dL/dZhat: what the row kernels produce
    zall = torch.cat([zh_i.detach(), zh_j.detach()]).to(dev).contiguous()
    inv_t, gscale = 2.0, 2.0 / (2 * nglob)
    splits = ops.ntxent_splits(nglob, b)
    assert splits > 1
    results = {}
    for mode in (1, splits):
        lse_all = torch.empty(2 * nglob, device=dev)
        total, parts = 0.0, []
        for r in range(WORLD):
            lse, pos = ops.ntxent_fwd(zall, nglob, b, r * b, inv_t, splits=mode)
            total += ops.ntxent_loss(lse, pos, 1.0 / (2 * nglob)).item()
            lse_all[r * b:(r + 1) * b] = lse[:b]
            lse_all[nglob + r * b:nglob + (r + 1) * b] = lse[b:]
        np.testing.assert_allclose(total, ref.item(), rtol=3e-6, err_msg=f"splits {mode}")
        for r in range(WORLD):
            dz = ops.ntxent_bwd(zall, lse_all, nglob, b, r * b, inv_t, gscale, splits=mode)
            close(dz[:b], zh_i.grad[r * b:(r + 1) * b], rtol=5e-4, what=f"splits {mode} rank {r} dzi")
            close(dz[b:], zh_j.grad[r * b:(r + 1) * b], rtol=5e-4, what=f"splits {mode} rank {r} dzj")
            parts.append(dz)
        results[mode] = (lse_all, torch.cat(parts))
    # the split kernels regroup the same sums: equal to the unsplit ones at rounding level
    close(results[splits][0], results[1][0], rtol=2e-6, what="lse split vs unsplit")
    close(results[splits][1], results[1][1], rtol=1e-4, what="dZ split vs unsplit")
    # the raw C entry points refuse a workspace that is too small / missing
    lse, pos = torch.empty(2 * b, device=dev), torch.empty(2 * b, device=dev)
    with pytest.raises(_lib.SsvError):
        _lib.call("ssv_ntxent_fwd_split", nglob, b, 0, d, _lib.ptr(zall), inv_t, _lib.ptr(lse), _lib.ptr(pos), splits, 0, 0, _lib.stream())
    ws = torch.empty(1024, dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.SsvError):
        _lib.call("ssv_ntxent_fwd_split", nglob, b, 0, d, _lib.ptr(zall), inv_t, _lib.ptr(lse), _lib.ptr(pos), splits, _lib.ptr(ws), 1024, _lib.stream())


def _build(dev):
    from ssv_amd.models import heads
    from ssv_amd.networks import resnet
    from ssv_amd.utils import train_utils
    torch.manual_seed(420)
    enc = resnet.resnet18(reduce_bottom_conv=True).to(dev)
    head = heads.SimclrProjectionHead(512, 128).to(dev)
    opt = train_utils.get_optimizer({"name": "sgd", "lr": 0.2, "weight_decay": 1e-4}, list(enc.parameters()) + list(head.parameters()))
    return enc, head, opt


def _arena_vs_params(flat, truth, other=None):
    """Relative l2 error against ``truth`` (a list in the oracle's parameter order) of every tensor of the arena-ordered flat vector (conv filters are
    OHWI in the arena) and, with ``other``, of that list too.  Tensors whose true gradient is analytically zero (a Linear bias in front of a
    BatchNorm) are compared absolutely and left out (tests/test_gpu_step.py::_check_grads)."""
    from ssv_amd.utils.train_utils import _ALIGN
    off, e_got, e_other = 0, [], []
    for i, p in enumerate(truth):
        n = p.numel()
        got = flat[off:off + n].double()
        got = got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)
        off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        if float(p.norm()) < 1e-5:
            assert float(got.abs().max()) < 1e-5, f"zero-gradient tensor {tuple(p.shape)} got {float(got.abs().max()):.2e}"
            continue
        e_got.append(float((got - p).norm()) / float(p.norm()))
        if other is not None:
            e_other.append(float((other[i].double() - p).norm()) / float(p.norm()))
    return e_got, e_other


def _oracle_pair():
    """The fp32 oracle and its fp64 twin (same fp32-drawn weights): ReLU mask flips make ANY two fp32 evaluations of this network differ at the
    1e-3 .. 1e-2 level in the early layers' gradients, so gradients are held to the fp32 CPU path's own distance from fp64 (tests/test_gpu_step.py)."""
    from test_gpu_step import _oracle64_like
    make = lambda: oracle.SimCLROracle("resnet18", True, 128, lr=0.2, weight_decay=1e-4)
    return make(), _oracle64_like(make)


def _rel(a, c):
    return float((a.double() - c.double()).norm() / c.double().norm())


def _check_against_fp64(flat, g32, g64, dz_hip, dz32, dz64):
    """TIGHT where no ReLU lies downstream: the gradient the loss hands to the heads (dz of this rank's rows - the quantity the sharding is about)
    and the projector's last layers, both as close to fp64 as the fp32 CPU path is.  FLIP-SIZE for the encoder: ONE ReLU input within rounding of
    zero that lands on the other side of the mask in a layer of n elements moves every earlier gradient by ~1/sqrt(n) - 5.5e-3 for the 32 K elements
    of layer4 at batch 16, and ~1 such flip per evaluation is expected there (profiles/r05_diag_rank_gradients.txt: the error sets in at exactly one
    ReLU and is the same with the loss gradient handed in from the oracle)."""
    for got, c32, c64 in zip(dz_hip, dz32, dz64):
        assert _rel(got.cpu(), c64) <= 3 * _rel(c32, c64) + 1e-6, (_rel(got.cpu(), c64), _rel(c32, c64))
    e_hip, e_cpu = _arena_vs_params(flat, g64, g32)
    # the tensors with NO ReLU decision in their gradient: fc2.weight, bn2.weight, bn2.bias (behind the head's ReLU) and bn1.weight (a flipped element has x_hat ~ 0, so it
    # drops out of sum g * x_hat).  bn1.BIAS is not one of them - its gradient is sum g * mask, and one flipped element of 16 rows moves it by ~1e-3 (rounds 3-5 took the
    # last FOUR tensors, bn1.bias among them: green only while no hidden unit of this batch sat within rounding of zero; round 6's arithmetic flips one).
    # Order of the non-zero-gradient tail: ..., fc1.weight, bn1.weight, bn1.bias, fc2.weight, bn2.weight, bn2.bias
    tail = (-5, -3, -2, -1)
    assert max(e_hip[i] for i in tail) <= 3 * max(e_cpu[i] for i in tail) + 1e-5, f"projector tail: hip {[e_hip[i] for i in tail]}, cpu fp32 {[e_cpu[i] for i in tail]}"
    assert max(e_hip) < 5e-2, f"worst per-tensor gradient error {max(e_hip):.2e} (flip size at this batch: ~1e-2)"


@pytest.mark.parametrize("k", [0, 5])
def test_resnet18_step_of_rank_k_of_8_matches_oracle_data_parallel_emulation(dev, k):
    from ssv_amd import distributed as hdist, nn as hnn, ops
    from ssv_amd.utils import losses
    b, nglob, ld = 16, 16 * WORLD, 128
    a1, a2 = seeded_randn(11, nglob, 3, 32, 32), seeded_randn(12, nglob, 3, 32, 32)
    sh = lambda t, r: t[r * b:(r + 1) * b].to(dev)
    enc, head, opt = _build(dev)
    # ---- the peers: every shard through the same weights, one after the other (BatchNorm statistics per shard and view)
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
    seen = []

    def gather(out, mine):
        if out.shape == (WORLD * 2 * b, ld):                   # the embedding all-gather: [zi_r ; zj_r] blocks in rank order
            seen.append("z")
            out.view(WORLD, 2 * b, ld).copy_(blocks)
            close(mine, blocks[k], rtol=1e-6, what="rank k's own embeddings are the ones its peers would receive")
            out.view(WORLD, 2 * b, ld)[k].copy_(mine)
        else:                                                  # the row log-sum-exps with the loss partial riding along
            seen.append("lse")
            assert out.shape == (WORLD, 2 * b + 4)
            out.copy_(packs)
            out[k].copy_(mine[0])
    prev = hdist.emulate_world(WORLD, k, gather=gather, reduce=lambda t: t)       # identity: the slab keeps rank k's own share of the SUM
    try:
        hdist.attach_grad_sync(opt, [enc, head])
        assert [n for n, _, _ in opt.grad_sync.buckets][:2] == ["ResNet.stage0", "ResNet.stage1"]
        loss_fn = losses.SimclrLoss(True, 0.5)
        with hnn.parallel_views(dev) as pv:
            with pv.view(0):
                z1 = head(enc(sh(a1, k)))
            with pv.view(1):
                z2 = head(enc(sh(a2, k)))
        z1.retain_grad(), z2.retain_grad()
        loss = loss_fn(z1, z2)
        opt.zero_grad()
        loss.backward()
        hnn.join_view_streams(dev)
        opt.grad_sync.finish()
        torch.cuda.synchronize()
    finally:
        hdist.restore_world(prev)
    assert seen == ["z", "lse"] and not hdist.is_on()
    # ---- oracle: all shards through one set of weights, loss on the concatenation; only rank k's embeddings carry the graph
    grads, dzs, ref32 = [], [], None
    for m, cast in zip(_oracle_pair(), (lambda t: t, lambda t: t.double())):
        mine = []

        def emb(x, r):
            z = m.embed(cast(x[r * b:(r + 1) * b]))
            if r != k:
                return z.detach()
            z.retain_grad()
            mine.append(z)
            return z
        ref = oracle.ntxent_loss(torch.cat([emb(a1, r) for r in range(WORLD)]), torch.cat([emb(a2, r) for r in range(WORLD)]), True, 0.5)
        ref.backward()
        grads.append([p.grad for p in m.params])
        dzs.append([z.grad for z in mine])
        ref32 = ref.item() if ref32 is None else ref32
    np.testing.assert_allclose(loss.item(), ref32, rtol=1e-5)
    _check_against_fp64(opt.arena.grad.cpu(), grads[0], grads[1], (z1.grad, z2.grad), dzs[0], dzs[1])


def test_replicated_peer_emulation_equals_the_oracle_on_the_repeated_batch(dev):
    """bench.py --emulate-world W: every emulated peer holds THIS rank's shard.  That job is exact: loss = oracle loss on the W x repeated
    batch, arena gradient after the emulated SUM = W x this rank's share = the oracle's full gradient of that batch."""
    from ssv_amd import distributed as hdist, nn as hnn
    from ssv_amd.utils import losses
    b = 16
    a1, a2 = seeded_randn(21, b, 3, 32, 32), seeded_randn(22, b, 3, 32, 32)
    enc, head, opt = _build(dev)
    prev = hdist.emulate_world(WORLD, 3)
    try:
        hdist.attach_grad_sync(opt, [enc, head])
        with hnn.parallel_views(dev) as pv:
            with pv.view(0):
                z1 = head(enc(a1.to(dev)))
            with pv.view(1):
                z2 = head(enc(a2.to(dev)))
        z1.retain_grad(), z2.retain_grad()
        loss = losses.SimclrLoss(True, 0.5)(z1, z2)
        opt.zero_grad()
        loss.backward()
        hnn.join_view_streams(dev)
        opt.grad_sync.finish()
        torch.cuda.synchronize()
    finally:
        hdist.restore_world(prev)
    grads, dzs, ref32 = [], [], None
    for m, cast in zip(_oracle_pair(), (lambda t: t, lambda t: t.double())):
        copies = [[m.embed(cast(a)) for _ in range(WORLD)] for a in (a1, a2)]       # per-shard BatchNorm: every copy sees the same statistics
        for zs in copies:
            zs[3].retain_grad()
        ref = oracle.ntxent_loss(torch.cat(copies[0]), torch.cat(copies[1]), True, 0.5)
        ref.backward()
        grads.append([p.grad for p in m.params])
        dzs.append([zs[3].grad for zs in copies])                                   # rank 3's rows of dL/dz
        ref32 = ref.item() if ref32 is None else ref32
    np.testing.assert_allclose(loss.item(), ref32, rtol=1e-5)
    _check_against_fp64(opt.arena.grad.cpu(), grads[0], grads[1], (z1.grad, z2.grad), dzs[0], dzs[1])


def test_vit_gradient_buckets_launch_from_the_backward_and_change_nothing(dev):
    """BASELINE config 5 on 8 GPUs: the ViT's gradient exchange in buckets (embedding + projection, encoder layers in runs of three, the head) launched from the backward
    pass - both student passes of a DINO step (global and local crops, two view slots) must have reported a bucket before it is reduced - against ONE reduction at
    step(): the same parameters, bit for bit (emulated world of 8: the reduction is x 8 on the exchange stream)."""
    import bench
    from ssv_amd import distributed as hdist
    enc = {"hidden_dim": 128, "embedding_dim": 16, "intermediate_dim": 256, "num_attention_heads": 2, "patch_size": 4,
           "num_local_patches": 4, "num_global_patches": 64, "num_encoder_layers": 6}
    mk = lambda seed, v, sz: seeded_randn(seed, 8, v, 3, sz, sz).to(dev)
    batch = {"global_1": mk(1, 2, 32), "global_2": mk(2, 2, 32), "local_1": mk(3, 4, 8), "local_2": mk(4, 4, 8)}
    saved = bench.BENCH_CFG["dino"]
    bench.BENCH_CFG["dino"] = dict(saved, encoder=enc, proj_head={"hidden_dim": 64, "proj_dim": 128})
    prev = hdist.emulate_world(WORLD, 0)
    try:
        runs = {}
        for bucketed in (True, False):
            step, _ = bench.build(dev, "dino")
            t = step.trainer
            hdist.detach_grad_sync(t.optim, t._sync_modules())
            hdist.attach_grad_sync(t.optim, t._sync_modules(), bucketed=bucketed)
            names = [b[0] for b in t.optim.grad_sync.buckets]
            launched = []
            inner = t.optim.grad_sync._launch
            t.optim.grad_sync._launch = lambda b, inner=inner, sync=t.optim.grad_sync: (launched.append(sync.buckets[b][0]), inner(b))[1]
            losses = [step(batch) for _ in range(2)]
            torch.cuda.synchronize()
            runs[bucketed] = (names, launched, losses, t.optim.arena.data.clone())
    finally:
        hdist.restore_world(prev)
        bench.BENCH_CFG["dino"] = saved
    names, launched, losses, params = runs[True]
    assert names == ["TransformerEncoder.stage0", "TransformerEncoder.stage1", "TransformerEncoder.stage2", "EncoderModel.rest"], names
    # the deepest run of layers first, the embedding last of the encoder; the head's parameters are the outer module's "rest": complete when its whole tape has run
    assert launched[:4] == ["TransformerEncoder.stage2", "TransformerEncoder.stage1", "TransformerEncoder.stage0", "EncoderModel.rest"], launched
    assert runs[False][0] == [] and runs[False][1] == []
    assert losses == runs[False][2] and torch.equal(params, runs[False][3])
