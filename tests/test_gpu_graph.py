"""GPU: the training step replayed as one HIP graph (ssv_amd.graph.StepGraph) against the eager step it was captured from - the reference's own regime
(configs/simclr.yaml: resnet18 reduce_bottom_conv on 32 x 32 images; models/simclr.py:86-95, models/byol.py:125-135, models/barlow.py:86-95).  Same kernels in the
same order on the same data: every step's loss and the parameters afterwards are BITWISE those of the eager run, through a learning-rate change (device memory: the same graph), a
ragged batch (eager fallback) and BYOL's per-step EMA between replays."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import seeded_randn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def _trainer(dev, algo):
    import bench
    step, _ = bench.build(dev, algo, arch="resnet18", reduce_bottom_conv=True, steps_per_epoch=50)
    return step.trainer


def _batches(dev, n, b=64):
    return [{"aug_1": seeded_randn(100 + 2 * i, b, 3, 32, 32).to(dev), "aug_2": seeded_randn(101 + 2 * i, b, 3, 32, 32).to(dev), "label": torch.zeros(b)} for i in range(n)]


@pytest.mark.parametrize("algo", ["simclr", "byol", "barlow"])
def test_replayed_steps_are_bitwise_the_eager_steps(dev, algo):
    from ssv_amd.graph import StepGraph
    batches = _batches(dev, 9)
    ragged = {k: v[:40] for k, v in _batches(dev, 1)[0].items()}
    runs = {}
    for mode in ("eager", "graph"):
        t = _trainer(dev, algo)
        sg = StepGraph(t, mode="1" if mode == "graph" else "0", graph_floors=False)          # the eager kernel selection: bitwise comparable
        losses = []
        for i, batch in enumerate(batches):
            if i == 5:                                           # the schedule moves the learning rate: four device floats, the same graph
                for g in t.optim.param_groups:
                    g["lr"] *= 0.5
            if i == 7:                                           # the ragged last batch of an epoch: runs eagerly in both
                losses.append(sg(ragged)["loss"])
                t._after_step(100)
            losses.append(sg(batch)["loss"])
            t._after_step(i)                                     # BYOL: tau schedule + EMA of the target between the steps, as in the train loop
        torch.cuda.synchronize()
        runs[mode] = (losses, t.optim.arena.data.clone(), t.optim.momentum_buffer.clone(), sg.describe())
    le, pe, me, _ = runs["eager"]
    lg, pg, mg, info = runs["graph"]
    assert info["disabled"] is None and info["graphs"] >= 1 and info["replays"] >= 5, info
    assert all(np.isfinite(le)) and le == lg, (le, lg)
    assert torch.equal(pe, pg) and torch.equal(me, mg)


def test_graph_kernel_selection_trains_like_the_eager_one(dev):
    """The shipped graph (ops.graph_dispatch: Winograd from 64 tiles / 64 channels on small images) runs OTHER kernels than the eager step - same mathematics:
    steps 0..2 (the same weights on both sides) within 1e-5 relative of the eager run's losses, the step after within the drift of two fp32 trajectories."""
    from ssv_amd import ops
    from ssv_amd.graph import StepGraph
    batches = _batches(dev, 6)
    runs = {}
    for mode in ("eager", "graph"):
        t = _trainer(dev, "simclr")
        sg = StepGraph(t, mode="1" if mode == "graph" else "0")
        losses = [sg(b)["loss"] for b in batches[:2]]
        prev, ops.DISPATCH = ops.DISPATCH, {}                                 # step 2: the capture in the graph run, one eager step in the other
        try:
            losses.append(sg(batches[2])["loss"])
            one_step = dict(ops.DISPATCH)
        finally:
            ops.DISPATCH = prev
        losses += [sg(b)["loss"] for b in batches[3:]]
        runs[mode] = (losses, one_step, sg.describe())
    (le, de, _), (lg, dg, info) = runs["eager"], runs["graph"]
    assert info["replays"] >= 3 and info["disabled"] is None
    assert sum(dg.values()) >= sum(de.values()) > 0, (de, dg)                # at least as many products on the Winograd forms under capture (lower floors)
    for a, b in zip(le[:3], lg[:3]):                                          # steps 0-1 are eager in both; step 2 evaluates the same weights with another selection
        assert abs(a - b) <= 1e-5 * abs(a), (le, lg)
    # from step 3 on the two runs carry weights that differ at rounding level: two fp32 evaluations of this trajectory at lr 0.2 drift apart like any two
    assert abs(le[3] - lg[3]) <= 5e-3 * abs(le[3]) and all(np.isfinite(lg)), (le, lg)


def test_auto_mode_graphs_small_images_only_and_unsafe_trainers_never(dev):
    from ssv_amd.graph import StepGraph
    t = _trainer(dev, "simclr")
    sg = StepGraph(t, mode="auto")
    for batch in _batches(dev, 5):
        sg(batch)
    assert sg.describe()["replays"] >= 2 and sg.describe()["disabled"] is None
    big = {"aug_1": seeded_randn(1, 4, 3, 96, 96).to(dev), "aug_2": seeded_randn(2, 4, 3, 96, 96).to(dev)}
    sg2 = StepGraph(_trainer(dev, "simclr"), mode="auto")
    for _ in range(4):
        sg2(big)
    assert sg2.describe()["replays"] == 0 and "GPU-bound" in sg2.describe()["disabled"]
    t.graph_safe = False
    sg3 = StepGraph(t, mode="1")
    sg3(_batches(dev, 1)[0])
    assert "graph_safe" in sg3.describe()["disabled"]
    # the trainer's own entry point: TwoViewTrainer.step
    t2 = _trainer(dev, "simclr")
    out = [t2.step(b)["loss"] for b in _batches(dev, 5)]
    assert all(np.isfinite(out)) and t2._step_graph.describe()["replays"] >= 2


def test_adamw_with_the_step_count_in_device_memory_is_the_by_value_update(dev):
    """ssv_adamw_counted (step count and bias corrections in device memory: what lets DINO's step be a graph) against ssv_adamw, three steps."""
    from ssv_amd import _lib
    n = 10_000
    p0, g = seeded_randn(1, n).to(dev), [seeded_randn(2 + i, n).to(dev) for i in range(3)]
    runs = []
    for counted in (False, True):
        p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        step_dev, bc = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(4, device=dev)
        for i in range(3):
            if counted:
                _lib.call("ssv_adamw_counted", n, _lib.ptr(p), _lib.ptr(g[i]), None, _lib.ptr(m), _lib.ptr(v), 1e-3, 0.9, 0.999, 1e-6, 0.04, _lib.ptr(step_dev), _lib.ptr(bc), 3.0, _lib.stream())
            else:
                _lib.call("ssv_adamw", n, _lib.ptr(p), _lib.ptr(g[i]), None, _lib.ptr(m), _lib.ptr(v), 1e-3, 0.9, 0.999, 1e-6, 0.04, i + 1, 3.0, _lib.stream())
        runs.append((p, m, v))
        if counted:
            assert int(step_dev.item()) == 3
    for a, b in zip(*runs):
        assert float((a - b).abs().max()) <= 1e-7 * float(b.abs().max())


def test_dino_step_replays_as_a_graph_through_an_epoch_schedule_change(dev):
    """DINO (models/dino.py:143-169) on a small ViT: AdamW's step count lives in device memory, the per-epoch scalars (temperatures; the weight decay is device memory) are part of the
    graph's key - the replayed steps are bitwise the eager ones, before and after `_after_epoch` moves the schedules (a new capture)."""
    import bench
    from ssv_amd.graph import StepGraph
    enc = {"hidden_dim": 128, "embedding_dim": 16, "intermediate_dim": 256, "num_attention_heads": 2, "patch_size": 4,
           "num_local_patches": 4, "num_global_patches": 64, "num_encoder_layers": 2}
    mk = lambda seed, v, sz: seeded_randn(seed, 8, v, 3, sz, sz).to(dev)
    batches = [{"global_1": mk(10 * i, 2, 32), "global_2": mk(10 * i + 1, 2, 32), "local_1": mk(10 * i + 2, 4, 8), "local_2": mk(10 * i + 3, 4, 8)} for i in range(8)]
    saved = bench.BENCH_CFG["dino"]
    bench.BENCH_CFG["dino"] = dict(saved, encoder=enc, proj_head={"hidden_dim": 64, "proj_dim": 128})
    try:
        runs = {}
        for mode in ("eager", "graph"):
            step, _ = bench.build(dev, "dino")
            t = step.trainer
            t.config.update(epochs=10, temp_warmup_epochs=3)
            sg = StepGraph(t, mode="1" if mode == "graph" else "0", graph_floors=False)
            losses = []
            for i, b in enumerate(batches):
                if i == 5:
                    t._after_epoch(1)                                  # teacher EMA, weight decay and teacher temperature move
                losses.append(sg(b)["loss"])
            torch.cuda.synchronize()
            runs[mode] = (losses, t.optim.arena.data.clone(), t.teacher_center.clone(), sg.describe())
    finally:
        bench.BENCH_CFG["dino"] = saved
    (le, pe, ce, _), (lg, pg, cg, info) = runs["eager"], runs["graph"]
    assert info["disabled"] is None and info["replays"] >= 4, info
    assert all(np.isfinite(le)) and le == lg, (le, lg)
    assert torch.equal(pe, pg) and torch.equal(ce, cg)


@pytest.mark.parametrize("algo", ["simsiam", "relic", "moco"])
def test_sibling_steps_replay_bitwise(dev, algo):
    """SimSiam (models/simsiam.py:122-132), ReLIC (models/relic.py:124-135: three forwards, the un-augmented image as third input) and MoCo (models/moco.py:113-125:
    the queue's write pointer lives in device memory, its push and the key encoder's EMA are part of the step) through the step graph."""
    from test_gpu_siblings import BASE, _bare
    from ssv_amd.graph import StepGraph
    from ssv_amd.models.moco import MoCo
    from ssv_amd.models.relic import ReLIC
    from ssv_amd.models.simsiam import SimSiam
    cls, cfg = {"moco": (MoCo, {**BASE, "proj_dim": 128, "queue_size": 40, "momentum": 0.999, "optimizer": {"name": "sgd", "lr": 0.03, "weight_decay": 1e-4},
                                "loss_fn": {"normalize": True, "temperature": 0.07}}),
                "simsiam": (SimSiam, {**BASE, "proj_dim": 256, "bottleneck_dim": 64, "optimizer": {"name": "sgd", "lr": 0.05, "weight_decay": 1e-4}}),
                "relic": (ReLIC, {**BASE, "proj_dim": 128, "tau": 0.996, "optimizer": {"name": "sgd", "lr": 0.2, "weight_decay": 1e-4},
                                  "loss_fn": {"normalize": True, "temperature": 1.0, "alpha": 0.5}})}[algo]
    batches = [{"img": seeded_randn(300 + 3 * i, 32, 3, 32, 32).to(dev), "aug_1": seeded_randn(301 + 3 * i, 32, 3, 32, 32).to(dev),
                "aug_2": seeded_randn(302 + 3 * i, 32, 3, 32, 32).to(dev)} for i in range(6)]
    runs = {}
    for mode in ("eager", "graph"):
        t = _bare(cls, dev, dict(cfg), loader_len=20)
        sg = StepGraph(t, mode="1" if mode == "graph" else "0", graph_floors=False)
        losses = []
        for i, b in enumerate(batches):
            losses.append(sg(b)["loss"])
            t._after_step(i)
        torch.cuda.synchronize()
        extra = (t.memory_bank.bank.clone(), t.memory_bank.ptr) if algo == "moco" else None
        runs[mode] = (losses, t.optim.arena.data.clone(), sg.describe(), extra)
    (le, pe, _, xe), (lg, pg, info, xg) = runs["eager"], runs["graph"]
    assert info["disabled"] is None and info["replays"] >= 3, info
    assert all(np.isfinite(le)) and le == lg and torch.equal(pe, pg), (le, lg)
    if algo == "moco":
        assert xe[1] == xg[1] == (6 * 32) % 40 and torch.equal(xe[0], xg[0])          # the queue and its pointer moved alike


def test_graph_survives_eager_work_between_replays_over_many_steps(dev):
    """60 steps with everything a real run interleaves: eager feature extraction (train-mode BatchNorm forward: running statistics move, the eager workspace and the
    weight caches are used) every 10 steps, a learning-rate change every 20 (re-capture), one ragged batch - parameters, momentum and BatchNorm running statistics
    end up BITWISE where the eager run's do."""
    from ssv_amd.graph import StepGraph
    batches = _batches(dev, 6)
    probe = seeded_randn(999, 48, 3, 32, 32).to(dev)
    runs = {}
    for mode in ("eager", "graph"):
        t = _trainer(dev, "simclr")
        sg = StepGraph(t, mode="1" if mode == "graph" else "0", graph_floors=False)
        feats = []
        for i in range(60):
            if i and i % 20 == 0:
                for g in t.optim.param_groups:
                    g["lr"] *= 0.7
            if i % 10 == 5:
                with torch.no_grad():
                    feats.append(t._features(probe).clone())
            if i == 33:
                sg({k: v[:24] for k, v in batches[0].items()})
            sg(batches[i % len(batches)])
        torch.cuda.synchronize()
        bn = torch.cat([b.flatten() for n, b in t.encoder.named_buffers() if "running" in n])
        runs[mode] = (t.optim.arena.data.clone(), t.optim.momentum_buffer.clone(), bn.clone(), torch.stack(feats), sg.describe())
    e, g = runs["eager"], runs["graph"]
    assert g[4]["graphs"] == 1 and g[4]["replays"] >= 50 and g[4]["disabled"] is None, g[4]
    for a, b, what in zip(e[:4], g[:4], ("parameters", "momentum", "BatchNorm running statistics", "features extracted between replays")):
        assert torch.equal(a, b), what


def test_a_per_step_learning_rate_schedule_is_served_by_one_graph(dev):
    """Learning rate, weight decay and momentum are device memory under capture (ssv_sgd_nesterov_dev; rounds 4-5 baked them into the graph and re-captured when they
    moved): a schedule that moves the learning rate EVERY step replays ONE graph, bitwise the eager run - and the optimizer's step count follows the executed steps."""
    from ssv_amd.graph import StepGraph
    b = _batches(dev, 1)[0]
    runs = {}
    for mode in ("0", "1"):
        t = _trainer(dev, "simclr")
        sg = StepGraph(t, mode=mode, graph_floors=False)
        losses = []
        for i in range(16):
            for g in t.optim.param_groups:
                g["lr"] = 0.2 * (1.0 - 0.01 * i)
                g["weight_decay"] = 1e-4 * (1.0 + 0.1 * i)
            losses.append(sg(b)["loss"])
        torch.cuda.synchronize()
        runs[mode] = (losses, t.optim.arena.data.clone(), t.optim._steps, sg.describe())
    (le, pe, ne, _), (lg, pg, ng, info) = runs["0"], runs["1"]
    assert info["captures"] == 1 and info["graphs"] == 1 and info["replays"] >= 13 and info["disabled"] is None, info
    assert all(np.isfinite(le)) and le == lg and torch.equal(pe, pg)
    assert ne == ng == 16


def test_a_graph_key_that_moves_every_step_sends_the_graph_back_to_the_eager_step(dev):
    """A scalar that still reaches a kernel as an ARGUMENT and moves every step (here: a trainer whose graph_key() does) would re-capture every step: after 8 captures with
    too few replays the wrapper gives up for good and the steps go on eagerly; at most MAX_LIVE graphs were ever live, the others retired."""
    from ssv_amd import graph
    t = _trainer(dev, "simclr")
    sg = graph.StepGraph(t, mode="1")
    b = _batches(dev, 1)[0]
    tick = [0]
    t.graph_key = lambda: (tick[0],)
    losses = []
    for i in range(16):
        tick[0] = i
        losses.append(sg(b)["loss"])
        assert len(sg.graphs) <= graph.MAX_LIVE
    info = sg.describe()
    assert info["captures"] == 8 and info["graphs"] == 0 and "changes too often" in info["disabled"], info
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_a_failing_capture_falls_back_to_an_untainted_eager_step(dev):
    """A capture that raises half way (here: in the loss) has already RECORDED the forward - the weight caches hold transposed / transformed / split filters that were never
    computed and live in the discarded graph's pool, and the optimizer counted a step.  The fallback must run the eager step on clean caches with the count rolled
    back: bitwise the step of a trainer that never tried to capture."""
    from ssv_amd import nn as hnn
    from ssv_amd.graph import StepGraph
    batches = _batches(dev, 4)
    runs = {}
    for mode in ("plain", "failing"):
        t = _trainer(dev, "simclr")
        sg = StepGraph(t, mode="1" if mode == "failing" else "0", graph_floors=False)
        if mode == "failing":
            real = t.loss_fn

            def boom(*a, **k):
                if hnn.capturing():
                    raise RuntimeError("injected failure under capture")
                return real(*a, **k)
            t.loss_fn = boom
        losses = [sg(b)["loss"] for b in batches]
        torch.cuda.synchronize()
        runs[mode] = (losses, t.optim.arena.data.clone(), t.optim._steps, sg.describe())
    (lp, pp, np_, _), (lf, pf, nf, info) = runs["plain"], runs["failing"]
    assert info["disabled"] is not None and "injected failure" in info["disabled"] and info["graphs"] == 0, info
    assert lp == lf and torch.equal(pp, pf) and np_ == nf == 4
