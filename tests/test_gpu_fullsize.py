"""GPU: the BASELINE workloads at (or near) their full per-GPU sizes, checked through size-independent properties - the CPU oracle
cannot run a ResNet-50 / ViT-S step at these batch sizes in test time, but the pieces that are cheap on the CPU can be checked
against what the GPU produced:
  * the loss the step returns equals the oracle's loss evaluated on the step's own embeddings (NT-Xent over 2x256 rows; the DINO
    loss over 20 crops/sample);
  * the two-stream schedule is bitwise identical to the sequential one and the step is bitwise repeatable;
  * one optimiser step moves every parameter tensor by a finite, non-zero amount and BatchNorm running statistics stay sane;
  * augmentation at full size: per-sample streams do not depend on the batch they are drawn in (shard invariance)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import vit as ovit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _views(dev, batch, size, step=0, ids=None):
    from ssv_amd.utils import augmentations
    cfg = {"color_jitter": {"brightness": 0.4, "contrast": 0.4, "saturation": 0.4, "hue": 0.1, "apply_prob": 0.8}, "random_gray": {"p": 0.2},
           "random_resized_crop": {"size": [size, size], "scale": [0.2, 1.0]}, "random_flip": None, "to_tensor": None,
           "normalize": {"mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}}
    tf = augmentations.get_transform(cfg)
    g = torch.Generator(device=dev).manual_seed(420)
    source = torch.randint(0, 256, (batch, size, size, 3), generator=g, device=dev, dtype=torch.uint8)
    ids = torch.arange(batch, device=dev, dtype=torch.int64) if ids is None else ids
    return tf, source, tf.apply(source, torch.arange(batch, device=dev, dtype=torch.int64), tf.draw(source, ids, step))


def test_simclr_resnet50_224_bs256_properties(dev):
    """BASELINE config 2 (SimCLR resnet50, synthetic 3x224x224, bs 256 on one GPU)."""
    from test_gpu_step import _Step
    b = 256
    _, _, views = _views(dev, b, 224)
    m1, m2 = _Step(dev, "resnet50", False), _Step(dev, "resnet50", False)
    before = m1.optim.arena.data.clone()
    loss1, z1, z2 = m1.step(views[0], views[1], dual=True)
    loss2, y1, y2 = m2.step(views[0], views[1], dual=False)
    assert loss1 == loss2 and torch.equal(z1, y1) and torch.equal(z2, y2)                    # two streams == one stream, bit for bit
    assert torch.equal(m1.optim.arena.data, m2.optim.arena.data)
    want = oracle.ntxent_loss(z1.cpu(), z2.cpu(), True, 0.5).item()                          # 512 x 512 Gram on the CPU
    np.testing.assert_allclose(loss1, want, rtol=1e-5)
    moved = (m1.optim.arena.data - before).abs()
    assert torch.isfinite(m1.optim.arena.data).all() and float(moved.max()) > 0
    off = 0
    for p in m1.params():
        n = p.numel()
        assert float(moved[off:off + n].max()) > 0, "a parameter tensor received no update"
        off += (n + 63) // 64 * 64
    for k, v in m1.state().items():
        if k.endswith("running_var"):
            assert float(v.min()) > 0 and torch.isfinite(v).all(), k
    np.testing.assert_allclose(z1.float().mean(0).abs().max().item(), 0.0, atol=1e-4)        # bn2 output: zero column mean (beta = 0)


def test_dino_vits16_bs128_loss_consistency_and_stream_equivalence(dev):
    """BASELINE config 5 (ViT-S/16, 2 x (2 x 224 + 8 x 96) crops per sample) at its full per-GPU batch of 128."""
    import bench
    from ssv_amd import nn as hnn
    from ssv_amd.utils import augmentations
    b = 128
    tf, source, _ = _views(dev, b, 224)
    mc = augmentations.MultiCrop({**bench.DINO_CROPS, "train_transforms": {
        "color_jitter": {"brightness": 0.4, "contrast": 0.4, "saturation": 0.4, "hue": 0.1, "apply_prob": 0.8}, "random_gray": {"p": 0.2},
        "random_resized_crop": {"size": [224, 224], "scale": [0.2, 1.0]}, "random_flip": None, "to_tensor": None,
        "normalize": {"mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}}})
    batch = mc(source, torch.arange(b, device=dev, dtype=torch.int64), 0)
    assert batch["global_1"].shape == (b, 2, 3, 224, 224) and batch["local_2"].shape == (b, 8, 3, 96, 96)
    losses, arenas = [], []
    for dual in (True, False):
        prev = hnn.set_view_streams(dual)
        try:
            step, _ = bench.build(dev, "dino")
            losses.append(step(batch))
        finally:
            hnn.set_view_streams(prev)
    assert losses[0] == losses[1]                                                             # stream schedule does not change a bit
    # loss consistency: rebuild the trainer, capture the student / teacher outputs of the step and score them with the oracle
    from ssv_amd.models.dino import DINO, _DinoLossFn
    captured = {}
    orig = _DinoLossFn.apply

    def spy(sg, sl, tg, center, bs, vg, vl, ts, tt):
        captured.update(sg=sg.detach().cpu(), sl=sl.detach().cpu(), tg=tg.detach().cpu(), c=center.detach().cpu().clone(), bs=bs, vg=vg, vl=vl, ts=ts, tt=tt)
        return orig(sg, sl, tg, center, bs, vg, vl, ts, tt)
    t = object.__new__(DINO)
    t.config = {"epochs": 1000, "scheduler": {"name": "cosine", "warmup_epochs": 10}, **bench.BENCH_CFG["dino"]}
    t.device, t.train_loader = dev, [None]
    torch.manual_seed(420)
    t._build("vit")
    t.loss_fn = spy
    got = t.train_step(batch)["loss"]
    assert got == losses[0]
    c = captured
    ng, nl, k = c["bs"] * c["vg"], c["bs"] * c["vl"], c["sg"].shape[1]
    s1 = torch.cat((c["sg"][:ng], c["sl"][:nl]), 0).view(c["bs"], -1, k)
    s2 = torch.cat((c["sg"][ng:], c["sl"][nl:]), 0).view(c["bs"], -1, k)
    t1, t2 = c["tg"][:ng].view(c["bs"], 2, k), c["tg"][ng:].view(c["bs"], 2, k)
    want = 0.5 * ovit.dino_loss(t1, s2, c["ts"], c["tt"], c["c"].view(1, -1)) + 0.5 * ovit.dino_loss(t2, s1, c["ts"], c["tt"], c["c"].view(1, -1))
    np.testing.assert_allclose(got, want.item(), rtol=1e-5)
    assert torch.isfinite(t.optim.arena.data).all() and torch.isfinite(t.teacher_center).all()


def test_full_size_augmentation_is_shard_invariant(dev):
    """bs 512 at 224x224: the views of samples 128..255 drawn inside the full batch equal those drawn as their own shard."""
    b = 512
    tf, source, full = _views(dev, b, 224, step=3)
    ids = torch.arange(128, 256, device=dev, dtype=torch.int64)
    part = tf.apply(source, ids, tf.draw(source, ids, 3))
    assert torch.equal(full[:, 128:256], part)
    assert torch.isfinite(full).all() and float(full.std()) > 0.5


def test_the_drivers_default_bench_command_carries_every_block_of_the_line():
    """`python bench.py` as the driver runs it (single GPU, no flags but a small batch / image so that the test takes a minute): ONE JSON line with the headline
    fields, `roofline`, `cpu_baseline`, `parity_gate` (teacher-forced pass, the batch-512 kernel selection recorded), `config1` (step graph beside the eager step),
    `config3_rank_emulation` (one rank of eight, emulated) and `other_configs.{byol, dino}` - every extra leg without an error string; and `--emulate-world 8`."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WANDB_MODE="disabled")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "SSV_DIST_FORCE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "32", "--size", "64", "--prof-steps", "1"],
                         env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "parity_gate", "config1", "config3_rank_emulation", "other_configs"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["dtype"] == "f32" and out["vs_baseline"] is None and out["value"] > 0
    gate = out["parity_gate"]
    assert gate["teacher_forced_pass"] and max(gate["loss_rel_err_teacher_forced"]) <= 1e-4, gate
    assert "winograd_launches_per_step" in gate["dispatch"]
    emu = out["config3_rank_emulation"]
    assert "error" not in emu and emu["emulated_world"] == 8 and emu["global_batch"] == 8 * 32 and emu["ms_per_step"] > 0 and emu["ntxent"]["splits"] >= 1, emu
    for name in ("byol", "dino"):
        leg = out["other_configs"][name]
        assert "error" not in leg and leg["value"] > 0 and isinstance(leg["pass"], bool), leg
        assert "error" not in leg["rank_of_8_emulation"] and leg["rank_of_8_emulation"]["ms_per_step"] > 0, leg["rank_of_8_emulation"]
    c1 = out["config1"]
    assert c1["gpu"]["step_graph"]["replays"] > 0 and c1["gpu"]["step_graph"]["disabled"] is None and c1["loss_step0"]["rel_err"] < 1e-4, c1
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--emulate-world", "8", "--steps", "2", "--warmup", "1", "--batch", "32", "--size", "64",
                          "--prof-steps", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["emulated_world"]["emulated_world"] == 8 and out["emulated_world"]["global_batch"] == 256 and len(out["emulated_world"]["gradient_buckets"]) >= 5
