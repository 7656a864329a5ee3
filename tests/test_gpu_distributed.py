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
        a1 = seeded_randn(1, B * WORLD, 3, 32, 32)[rank * B:(rank + 1) * B].to(dev)
        a2 = seeded_randn(2, B * WORLD, 3, 32, 32)[rank * B:(rank + 1) * B].to(dev)

        def one_step(bucketed):
            torch.manual_seed(420)
            enc = resnet.resnet18(reduce_bottom_conv=True).to(dev)
            head = heads.SimclrProjectionHead(512, 128).to(dev)
            opt = train_utils.get_optimizer({"name": "sgd", "lr": 0.2, "weight_decay": 1e-4}, list(enc.parameters()) + list(head.parameters()))
            hdist.attach_grad_sync(opt, [enc, head], bucketed=bucketed)
            names = [b[0] for b in opt.grad_sync.buckets]
            assert names == (["ResNet.stage0", "ResNet.stage1", "ResNet.stage2", "ResNet.stage3", "ResNet.stage4", "SimclrProjectionHead.rest"] if bucketed else []), names
            launched = []
            inner = opt.grad_sync._launch
            opt.grad_sync._launch = lambda b: (launched.append(opt.grad_sync.buckets[b][0]), inner(b))[1]
            loss_fn = losses.SimclrLoss(True, 0.5)
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
            return loss, opt, launched
        # the gradient exchange per bucket, launched from the backward pass (head first, then layer4 ... stem), against ONE call at step()
        loss_s, opt_s, launched_s = one_step(False)
        loss, opt, launched = one_step(True)
        assert launched_s == [] and launched == ["SimclrProjectionHead.rest", "ResNet.stage4", "ResNet.stage3", "ResNet.stage2", "ResNet.stage1", "ResNet.stage0"], launched
        assert loss.item() == loss_s.item() and torch.equal(opt.arena.data, opt_s.arena.data), "bucketed gradient exchange is not bitwise the single call"
        extra = _sharded_loss_checks(rank, dev, losses)
        dino = _dino_step(rank, dev)
        q.put((rank, "ok" if extra is None else extra, loss.item(), opt.arena.data.cpu().numpy(), dino))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None, None, None))


DINO_ENC = {"hidden_dim": 128, "embedding_dim": 16, "intermediate_dim": 256, "num_attention_heads": 2, "patch_size": 4,
            "num_local_patches": 4, "num_global_patches": 16, "num_encoder_layers": 2}
DINO_HEAD = {"hidden_dim": 64, "proj_dim": 128}
DINO_BS, DINO_VL = 3, 2


def _dino_batch(rank):
    sl = slice(rank * DINO_BS, (rank + 1) * DINO_BS)
    return {"global_1": seeded_randn(70, WORLD * DINO_BS, 2, 3, 16, 16)[sl], "global_2": seeded_randn(71, WORLD * DINO_BS, 2, 3, 16, 16)[sl],
            "local_1": seeded_randn(72, WORLD * DINO_BS, DINO_VL, 3, 8, 8)[sl], "local_2": seeded_randn(73, WORLD * DINO_BS, DINO_VL, 3, 8, 8)[sl]}


def _dino_step(rank, dev):
    """One data-parallel DINO step with the real kernels: returns (loss, parameters after the update, centre)."""
    from ssv_amd import distributed as hdist
    from ssv_amd.models.dino import DINO
    from ssv_amd.utils import train_utils
    t = object.__new__(DINO)
    t.config = {"epochs": 100, "gradient_clip": 0.05, "encoder": DINO_ENC, "proj_head": DINO_HEAD,
                "optimizer": {"name": "adamw", "lr": 1e-3, "epsilon": 1e-6, "weight_decay": 0.04}, "scheduler": {"name": "cosine", "warmup_epochs": 0}}
    t.device, t.train_loader = dev, [None]
    torch.manual_seed(420)
    t._build("vit")
    hdist.attach_grad_sync(t.optim, t._sync_modules())
    loss = t.train_step(_dino_batch(rank))["loss"]
    torch.cuda.synchronize()
    return loss, t.optim.arena.data.cpu().numpy(), t.teacher_center.cpu().numpy()


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
    # ReLIC (NT-Xent + the batch-wide invariance term) and MoCo (per-sample loss against a replicated queue) over the split batch
    from oracle import siblings as osib
    zs = [seeded_randn(50 + k, n, 64) for k in range(3)]
    ref_in = [z.clone().requires_grad_() for z in zs]
    ref = osib.relic_loss(*ref_in, normalize=True, temperature=0.5, alpha=0.7)
    ref.backward()
    loc = [z[sl].to(dev).requires_grad_() for z in zs]
    loss = losses.RelicLoss(True, 0.5, 0.7)(*loc)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-5)
    for k in range(3):
        np.testing.assert_allclose(loc[k].grad.cpu().numpy(), ref_in[k].grad[sl].numpy(), rtol=2e-3, atol=2e-6)
    q, kk, bank = seeded_randn(60, n, 64), seeded_randn(61, n, 64), torch.nn.functional.normalize(seeded_randn(62, 80, 64), dim=1)
    qr = q.clone().requires_grad_()
    ref = osib.moco_loss(qr, kk, bank, normalize=True, temperature=0.2)
    ref.backward()
    ql = q[sl].to(dev).requires_grad_()
    loss = losses.MocoLoss(True, 0.2)(ql, kk[sl].to(dev), bank.to(dev).contiguous(), 80)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-5)
    np.testing.assert_allclose(ql.grad.cpu().numpy(), qr.grad[sl].numpy(), rtol=2e-3, atol=2e-7)
    # the queue of every rank receives the keys of ALL ranks, in rank order
    from ssv_amd.models.moco import MemoryBank
    mb = MemoryBank(96, 64, dev)
    mb.add_batch(kk[sl].to(dev))
    want = torch.nn.functional.normalize(kk, dim=1)
    np.testing.assert_allclose(mb.get_vectors()[:n].cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-7)
    assert mb.ptr == n % 96
    return None


def _check_dino(d0, d1):
    """Both ranks hold the same loss / parameters / centre, and they equal the oracle's emulation: every shard scored on its own
    (loss = mean of the shard losses), gradients averaged over shards BEFORE the clamp, centre = EMA of the global teacher mean."""
    from oracle import vit as ovit
    assert d0[0] == d1[0]
    np.testing.assert_array_equal(d0[1], d1[1])
    np.testing.assert_array_equal(d0[2], d1[2])
    o = ovit.DinoOracle(DINO_ENC, DINO_HEAD, lr=1e-3, weight_decay=0.04, warmup_epochs=0, clip=0.05)
    grads, losses, tmeans = None, [], []
    for r in range(WORLD):
        b = _dino_batch(r)
        shard = ovit.DinoOracle(DINO_ENC, DINO_HEAD, lr=1e-3, weight_decay=0.04, warmup_epochs=0, clip=None)
        out = shard.train_step(b["global_1"], b["global_2"], b["local_1"], b["local_2"], return_outputs=True)
        losses.append(out["loss"])
        tmeans.append(torch.cat((out["teacher_1"].reshape(-1, DINO_HEAD["proj_dim"]), out["teacher_2"].reshape(-1, DINO_HEAD["proj_dim"])), 0).mean(0))
        grads = shard.last_grads if grads is None else {k: grads[k] + v for k, v in shard.last_grads.items()}
    np.testing.assert_allclose(d0[0], np.mean(losses), rtol=1e-5)
    mean_grads = {k: v / WORLD for k, v in grads.items()}
    ovit.adamw_step(o.student, mean_grads, o.opt_state, o.lr, o.weight_decay, o.eps, clip=0.05)
    from ssv_amd.utils.train_utils import _ALIGN
    off = 0
    for k, ref in o.student.items():
        n = ref.numel()
        got = torch.from_numpy(d0[1][off:off + n]).view(ref.shape)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-3, atol=2e-5, err_msg=k)       # lr-sized AdamW moves: sign(g) * lr at step 1
        off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
    center = 0.9 * o.center + 0.1 * torch.stack(tmeans).mean(0)
    np.testing.assert_allclose(d0[2], center.numpy(), rtol=1e-4, atol=1e-5)


def test_two_ranks_match_oracle_data_parallel_emulation():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(WORLD)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for rank, msg, *_ in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"
    (_, _, l0, p0, d0), (_, _, l1, p1, d1) = res
    _check_dino(d0, d1)
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
    # (which ReLUs flip at rounding level depends on the kernels' summation order: medians of 5e-5 .. 3e-4 were measured across kernel states)
    assert np.median(errs) < 1e-3 and max(errs) < 3e-2, f"median {np.median(errs):.2e}, worst {max(errs):.2e}"


# ---------------------------------------------------------------------------------------------------------------------
_RCCL_WORLD1 = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["SSV_ROOT"])
from ssv_amd import distributed as hdist
from ssv_amd.models import heads
from ssv_amd.networks import resnet
from ssv_amd.utils import losses, train_utils
from conftest import seeded_randn

dev = torch.device("cuda", 0)
a1, a2 = seeded_randn(1, 16, 3, 32, 32).to(dev), seeded_randn(2, 16, 3, 32, 32).to(dev)


def step():
    torch.manual_seed(420)
    enc, head = resnet.resnet18(reduce_bottom_conv=True).to(dev), heads.SimclrProjectionHead(512, 128).to(dev)
    opt = train_utils.get_optimizer({"name": "sgd", "lr": 0.2, "weight_decay": 1e-4}, list(enc.parameters()) + list(head.parameters()))
    hdist.attach_grad_sync(opt)
    loss = losses.SimclrLoss(True, 0.5)(head(enc(a1)), head(enc(a2)))
    opt.zero_grad(); loss.backward(); opt.step()
    zi, zj = seeded_randn(5, 16, 64).to(dev).requires_grad_(), seeded_randn(6, 16, 64).to(dev).requires_grad_()
    bl = losses.BarlowLoss(True, 0.005)(zi, zj); bl.backward()
    torch.cuda.synchronize()
    return loss.item(), opt.arena.data.cpu().numpy(), bl.item(), zi.grad.cpu().numpy(), opt.grad_sync is not None


plain = step()                                     # no process group: the collective helpers are the identity
assert not hdist.is_on() and not plain[4]
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["SSV_PORT"], SSV_DIST_FORCE="1")
hdist.init_from_env(backend="nccl")
import torch.distributed as dist
assert hdist.is_on() and dist.get_backend() == "nccl" and hdist.world_size() == 1
# the raw helpers on RCCL: in-place all_gather_into_tensor with the clone for the aliasing input, SUM all-reduce, broadcast
buf = torch.arange(12, dtype=torch.float32, device=dev).view(6, 2).clone()
want = buf.clone()
assert torch.equal(hdist.all_gather_rows(buf, 6), want)
t = torch.full((1 << 20,), 3.0, device=dev)
assert torch.equal(hdist.all_reduce_sum(t), torch.full_like(t, 3.0))
assert torch.equal(hdist.broadcast_parameters(t), torch.full_like(t, 3.0))
assert hdist.broadcast_object("x") == "x"
hdist.barrier()
forced = step()                                    # all-gather of z, LSE exchange, loss all-reduce, gradient all-reduce - all on RCCL
assert forced[4]
torch.cuda.synchronize()
dist.destroy_process_group()
print(json.dumps({"loss": [plain[0], forced[0]], "barlow": [plain[2], forced[2]],
                  "params_equal": bool(np.array_equal(plain[1], forced[1])), "dz_equal": bool(np.array_equal(plain[3], forced[3]))}))
"""


def test_rccl_collectives_run_on_hardware_with_a_world_of_one(tmp_path):
    """backend "nccl" (= RCCL) with WORLD_SIZE=1 and SSV_DIST_FORCE=1: all_gather_into_tensor / all_reduce / broadcast of
    distributed.py execute on the GPU, and a full SimCLR step + a Barlow loss routed through them are BITWISE the step without a
    process group (one rank: the global batch is the local batch)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_world1.py"
    script.write_text(_RCCL_WORLD1)
    env = dict(os.environ, SSV_ROOT=root, SSV_PORT=str(_free_port()), PYTHONPATH=os.pathsep.join([root, os.path.join(root, "tests")]),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SSV_DIST_BACKEND", "SSV_DIST_FORCE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])      # RCCL prints its own banner lines
    assert out["loss"][0] == out["loss"][1] and out["barlow"][0] == out["barlow"][1], out
    assert out["params_equal"] and out["dz_equal"], out


def _cli_worker(rank, port, workdir, cfg_path, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank),
                          SSV_DIST_BACKEND="gloo", WANDB_MODE="disabled")
        os.chdir(workdir)
        import torch.distributed as dist
        from ssv_amd import main as cli
        from ssv_amd.utils import data_utils
        seen = []
        make = data_utils.GpuTwoViewLoader._make

        def spy(self, idx, step):
            seen.append((step, idx.cpu().tolist()))
            return make(self, idx, step)
        data_utils.GpuTwoViewLoader._make = spy
        model = cli.main(["-c", cfg_path, "-a", "simclr", "-m", "resnet18", "-t", "train"])        # default --output: rank 0's timestamp
        torch.cuda.synchronize()
        q.put((rank, "ok", seen, model.output_dir, model.optim.arena.data.cpu().numpy(), len(model.train_loader)))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None, None, None, None))


def test_cli_two_ranks_train_on_disjoint_shards(tmp_path):
    """`torch.distributed.run --nproc-per-node 2 main.py ...` semantics (two ranks, gloo, one GPU): the loaders hand the ranks
    disjoint halves of every global batch of the shared permutation, only rank 0 writes the run directory, replicas stay identical."""
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "self-supervised-vision_amd", "configs", "simclr.yaml")))
    cfg["epochs"], cfg["eval_every"] = 2, 1
    cfg["data"]["batch_size"] = 16                                       # per GPU: global batch 32
    cfg["data"]["synthetic"] = {"num_train": 72, "num_test": 40, "image_size": [32, 32], "num_classes": 10}     # 72 = 2 global batches + 8
    cfg["linear_eval"]["epochs"] = 2
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.dump(cfg, sort_keys=False))
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, port, str(tmp_path), str(path), q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(WORLD)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for rank, msg, *_ in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"
    (_, _, s0, d0, p0, n0), (_, _, s1, d1, p1, n1) = res
    assert n0 == n1 == 3 and len(s0) == len(s1) == 6                    # 3 steps per epoch on each rank, 2 epochs
    gen = torch.Generator().manual_seed(420)
    for epoch in range(2):
        order = torch.randperm(72, generator=gen).tolist()
        for st in range(3):
            (ga, a), (gb, b) = s0[3 * epoch + st], s1[3 * epoch + st]
            assert ga == gb == 3 * epoch + st                            # the augmentation stream key is the GLOBAL step
            assert not set(a) & set(b) and a + b == order[32 * st:32 * st + len(a) + len(b)]
            assert len(a) == len(b) == (16 if st < 2 else 4)
    assert d0 == d1                                                      # rank 0's run name on every rank
    np.testing.assert_array_equal(p0, p1)                                # identical replicas after training on different shards
    out = tmp_path / d0
    log = (out / "trainlogs.txt").read_text()
    assert log.count("[TRAIN] Epoch    1/   2") == 1 and (out / "best_model.pt").exists()      # one writer


def _plain_env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", WANDB_MODE="disabled", **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "SSV_DIST_FORCE"):
        env.pop(k, None)
    return env


def test_bench_gpus_2_starts_its_two_ranks_itself():
    """The driver's invocation for N > 1, `python bench.py --gpus 2 ...` with NO launcher around it: the parent (which never imports torch,
    tests/test_launch_cpu.py) starts two ranks as child processes and rank 0's JSON line comes back through it.  gloo, because RCCL refuses
    two ranks on the test box's one GPU; on an N-GPU node the same path runs with the default backend "nccl" (= RCCL)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
                          "--no-cpu-baseline", "--prof-steps", "0"], env=_plain_env(SSV_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                                        # ONE line, from rank 0
    out = json.loads(lines[0])
    d = out["distributed"]
    assert out["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo" and d["launched_by"].startswith("bench.py")
    assert [x["rank"] for x in d["devices"]] == [0, 1] and len({x["pid"] for x in d["devices"]}) == 2
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 32 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-3     # whole-job images/s over the max-over-ranks time
    assert d["gradient_buckets"] and d["comm_ms_per_step"] > 0
    assert 0 < out["config"]["last_loss"] < 10


def test_main_ssv_gpus_2_starts_two_ranks(tmp_path):
    """`SSV_GPUS=2 python main.py <the reference's flags>`: two ranks train on disjoint shards, one run directory, one writer."""
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "self-supervised-vision_amd", "configs", "simclr.yaml")))
    cfg["epochs"], cfg["eval_every"] = 1, 1
    cfg["data"]["batch_size"] = 16
    cfg["data"]["synthetic"] = {"num_train": 64, "num_test": 40, "image_size": [32, 32], "num_classes": 10}
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.dump(cfg, sort_keys=False))
    res = subprocess.run([sys.executable, os.path.join(root, "main.py"), "-c", str(path), "-a", "simclr", "-m", "resnet18", "-t", "train", "-o", "run"],
                         env=_plain_env(SSV_GPUS="2", SSV_DIST_BACKEND="gloo"), cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = tmp_path / "outputs" / "simclr" / "resnet18" / "run"
    log = (out / "trainlogs.txt").read_text()
    assert log.count("[TRAIN] Epoch    1/   1") == 1 and (out / "best_model.pt").exists()
