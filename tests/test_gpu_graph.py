"""GPU: the training step replayed as one HIP graph (ssv_amd.graph.StepGraph) against the eager step it was captured from - the reference's own regime
(configs/simclr.yaml: resnet18 reduce_bottom_conv on 32 x 32 images; models/simclr.py:86-95, models/byol.py:125-135, models/barlow.py:86-95).  Same kernels in the
same order on the same data: every step's loss and the parameters afterwards are BITWISE those of the eager run, through a learning-rate change (new capture), a
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
            if i == 5:                                           # the schedule moves the learning rate: a new graph
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
    step 0..2 losses within 1e-4 relative of the eager run's (two fp32 evaluations of one trajectory), and the selection really differs."""
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
    assert sum(dg.values()) > sum(de.values()), (de, dg)                     # more products on the Winograd forms under capture
    for a, b in zip(le[:4], lg[:4]):                                          # steps 0-1 are eager in both; 2-3 differ by the selection only
        assert abs(a - b) <= 1e-4 * abs(a), (le, lg)
    assert all(np.isfinite(lg))


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
