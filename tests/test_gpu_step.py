"""GPU parity of blocks, networks and whole training steps against the oracle and the fixtures
generated from the reference (tests/golden).  Tolerances are stated per check; the north-star bar
is per-step loss within 1e-4 relative and projected features within 1e-4 absolute."""
import numpy as np
import pytest
import torch

import oracle
from conftest import seeded_randn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def close(got, ref, rtol=1e-4, atol=None, what=""):
    got, ref = got.detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
    scale = float(ref.abs().max()) + 1e-30
    atol = 2e-5 * scale if atol is None else atol
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{what}: max err {float(err.max()):.3e} (scale {scale:.3e}), {int(bad.sum())}/{bad.numel()} out of tolerance"


def rel_l2(got, ref):
    got, ref = got.detach().cpu().double().flatten(), torch.as_tensor(ref).detach().cpu().double().flatten()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def _load_block(blk, g, prefix):
    sd = {k: torch.tensor(g[f"{prefix}_sd_{k}"]) for k in blk.state_dict().keys()}
    blk.load_state_dict(sd)


def test_bottleneck_block_matches_reference(dev, golden):
    from ssv_amd.networks import resnet
    g = golden["block_level"]
    blk = resnet.Bottleneck(64, 64, stride=2, downsample=resnet._Downsample(64, 256, 2))
    _load_block(blk, g, "bottleneck")
    blk = blk.to(dev)
    x = seeded_randn(71, 2, 64, 8, 8).permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()   # blocks take NHWC
    y = blk(x)
    y.backward(seeded_randn(72, 2, 256, 4, 4).permute(0, 2, 3, 1).contiguous().to(dev))
    close(y.permute(0, 3, 1, 2), g["bottleneck_y"], what="bottleneck y")
    close(x.grad.permute(0, 3, 1, 2), g["bottleneck_dx"], rtol=3e-4, what="bottleneck dx")
    for k, p in blk.named_parameters():
        assert rel_l2(p.grad.contiguous(), g[f"bottleneck_grad_{k}"]) < 2e-5, k
    # the fixture's state_dict was taken AFTER the reference forward (running stats updated once from 0 / 1);
    # loading it and running our forward on the same input is the second update with the same batch statistics
    for k, v in blk.state_dict().items():
        ref = torch.tensor(g[f"bottleneck_sd_{k}"])
        if k.endswith("running_mean"):
            close(v, 1.9 * ref, what=k)
        elif k.endswith("running_var"):
            close(v, 1.9 * ref - 0.9, what=k)
        elif k.endswith("num_batches_tracked"):
            assert int(v) == int(ref) + 1


def test_basic_block_matches_reference(dev, golden):
    from ssv_amd.networks import resnet
    g = golden["block_level"]
    blk = resnet.BasicBlock(32, 32)
    _load_block(blk, g, "basic")
    blk = blk.to(dev)
    x = seeded_randn(81, 2, 32, 6, 6).permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()
    y = blk(x)
    y.backward(seeded_randn(82, 2, 32, 6, 6).permute(0, 2, 3, 1).contiguous().to(dev))
    close(y.permute(0, 3, 1, 2), g["basic_y"], what="basic y")
    close(x.grad.permute(0, 3, 1, 2), g["basic_dx"], rtol=3e-4, what="basic dx")
    for k, p in blk.named_parameters():
        assert rel_l2(p.grad.contiguous(), g[f"basic_grad_{k}"]) < 2e-5, k


def _check_grads(m, o, flip_tol=2e-2):
    """Per-tensor gradient parity.  Two effects are NOT kernel errors and are allowed for explicitly:
      * a Linear bias that feeds a BatchNorm has an analytically ZERO gradient - both sides hold rounding
        noise there, so it is compared absolutely against the weight-gradient scale;
      * ReLU is discontinuous: an activation within ~1e-6 of zero can sit on different sides of the mask in
        two fp32 evaluations; ONE such flip in a layer of n elements moves that layer's (and every earlier
        layer's) gradient by ~1/sqrt(n) ~ 1e-3 relative.  (Same effect between the fp32 and an fp64 CPU run:
        tools/diag_grads.py.)  Hence this helper only bounds the error by the flip size; the TIGHT statement
        is test_gradients_as_close_to_fp64_truth_as_the_cpu_path below; the multi-step 1e-4 statement is tests/test_gpu_trajectories.py."""
    errs = []
    for p, go, off in zip(m.params(), o.last_grads, m.optim.arena.offsets):
        got = m.grads[off:off + p.numel()]
        got = got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)
        if float(go.norm()) < 1e-5:
            assert float(got.abs().max()) < 1e-5, f"zero-gradient tensor {tuple(p.shape)} got {float(got.abs().max()):.2e}"
            continue
        errs.append(rel_l2(got, go))
    assert max(errs) < flip_tol, f"worst per-tensor gradient error {max(errs):.2e}"
    return errs


class _Step:
    """The reference train_step body (models/simclr.py:86-95) on the product's components."""

    def __init__(self, dev, arch, rbc, proj_dim=128, lr=2.0, wd=1e-4, warmup=10, normalize=True, temperature=0.5):
        from ssv_amd.models import heads
        from ssv_amd.networks import resnet
        from ssv_amd.utils import losses, train_utils
        torch.manual_seed(420)
        self.encoder = getattr(resnet, arch)(**({"reduce_bottom_conv": True} if rbc else {})).to(dev)
        self.proj_head = heads.SimclrProjectionHead(self.encoder.out_dim, proj_dim).to(dev)
        self.optim = train_utils.get_optimizer({"name": "sgd", "lr": lr, "weight_decay": wd},
                                               list(self.encoder.parameters()) + list(self.proj_head.parameters()))
        train_utils.get_scheduler({"name": "cosine", "warmup_epochs": warmup, "epochs": 1000}, self.optim)
        self.loss_fn = losses.SimclrLoss(normalize, temperature)
        self.dev = dev

    def step(self, a1, a2, dual=False):
        from ssv_amd import nn as hnn
        a1, a2 = a1.to(self.dev), a2.to(self.dev)
        with hnn.parallel_views(self.dev, enabled=dual) as pv:
            with pv.view(0):
                z1 = self.proj_head(self.encoder(a1))
            with pv.view(1):
                z2 = self.proj_head(self.encoder(a2))
        loss = self.loss_fn(z1, z2)
        self.optim.zero_grad()
        loss.backward()
        from ssv_amd import nn as hnn
        hnn.join_view_streams(self.dev)
        self.grads = self.optim.arena.grad + self.optim.arena.grad_alt       # test-side read-out of the two slabs
        self.optim.step()
        return loss.item(), z1.detach(), z2.detach()

    def params(self):
        return list(self.encoder.parameters()) + list(self.proj_head.parameters())

    def state(self):
        return {**{"encoder." + k: v for k, v in self.encoder.state_dict().items()},
                **{"proj_head." + k: v for k, v in self.proj_head.state_dict().items()}}


def test_simclr_r18_steps_match_reference_and_oracle(dev, golden):
    """BASELINE config 1 shape (resnet18 rbc, 32x32, bs 64, configs/simclr.yaml hyper-parameters)."""
    g = golden["step_level"]
    m = _Step(dev, "resnet18", True)
    make = lambda: oracle.SimCLROracle("resnet18", True, 128, lr=float(g["simclr_r18_lr"]), weight_decay=1e-4)
    o, o64 = make(), _oracle64_like(make)
    assert abs(m.optim.param_groups[0]["lr"] - float(g["simclr_r18_lr"])) < 1e-15
    losses = []
    for s in range(3):
        a1, a2 = seeded_randn(100 + 2 * s, 64, 3, 32, 32), seeded_randn(101 + 2 * s, 64, 3, 32, 32)
        loss, z1, z2 = m.step(a1, a2)
        ref = o.train_step(a1, a2, return_z=(s == 0))
        losses.append(loss)
        if s == 0:
            np.testing.assert_allclose(loss, g["simclr_r18_losses"][0], rtol=1e-5)
            close(z1, g["simclr_r18_z1"], rtol=1e-4, atol=1e-4, what="z_1")          # north-star: |dz| <= 1e-4
            close(z2, g["simclr_r18_z2"], rtol=1e-4, atol=1e-4, what="z_2")
            # gradients of step 0, tensor by tensor (relative l2 vs the oracle)
            _check_grads(m, o)
            for k, ref_sum in zip(g["simclr_r18_after1_keys"], g["simclr_r18_after1_sums"]):
                got = oracle.tensor_checksum(m.state()[str(k)].contiguous())
                scale = float(np.sqrt(ref_sum[1])) + 1e-6
                # plain sums cancel; a ReLU flip moves a gradient by ~3e-3 relative and the first update is
                # 0.38*g, so the sum is only pinned to a few 1e-2 of the tensor norm (which flips occur depends on the summation
                # order inside the kernels), the sum of squares tightly
                np.testing.assert_allclose(got[0], ref_sum[0], rtol=1e-4, atol=3e-2 * scale, err_msg=str(k))
                # BN biases start at 0, so after one step they ARE the (flip-noisy) gradient: flip-size tolerance there
                # (weights: 1e-4, except the stem, whose gradient is the largest relative to its weights - one more or one fewer flipped ReLU
                # downstream moves |w'|^2 by 1e-4 .. 2e-4: measured 1.75e-4 with the row-taps stem kernels, 6e-5 with the 4-channel ones)
                np.testing.assert_allclose(got[1], ref_sum[1], rtol=2e-2 if str(k).endswith(".bias") else (5e-4 if str(k) == "encoder.conv1.weight" else 1e-4),
                                           atol=1e-9, err_msg=str(k))
        # Training at lr 0.2 amplifies rounding (and ReLU-flip) differences of the first updates: by step 2 the
        # fp32 CPU path itself is ~9e-4 away from an fp64 evaluation.  So every step is bounded by the CPU
        # path's own distance to the fp64 truth, and steps 0-1 additionally by the north-star 1e-4.
        l64 = o64.train_step(a1.double(), a2.double())["loss"]
        # the CPU path's own distance to fp64 is ONE sample of a chaotic quantity (it moves with the thread count); every rounding-
        # level change inside a kernel (k-loop order, how BatchNorm partials are grouped) flips a different set of ReLUs at step 0
        # and lands somewhere else at step 2: 8e-4 (CPU fp32), 4e-3 and 8e-3 were measured for three kernel states.  Steps 0-1 carry
        # the parity bar; from step 2 on only the size class is checked.
        # Step 0 is a pure function of the inputs: 1e-5.  Step 1 already sees one update at lr 0.2: a rounding-level (1e-7) difference in
        # the step-0 gradients - which FMA contractions the compiler picks in an epilogue is enough - comes back as ~1e-4 in this loss
        # (measured 5e-5, 1.2e-4 and 3.2e-4 for three builds of the same arithmetic).  The 1e-4 statement is made where it is well
        # posed: tests/test_gpu_trajectories.py (steps 0-2 at bs 128, lr / 100, and the fp32-ensemble envelope at this lr).
        slack = (2e-2 if s >= 2 else (1e-3 if s == 1 else 1e-5)) * abs(l64)
        assert abs(loss - l64) <= 3 * abs(ref["loss"] - l64) + slack, f"step {s}: hip {loss} cpu32 {ref['loss']} cpu64 {l64}"
        if s < 2:
            np.testing.assert_allclose(loss, ref["loss"], rtol=1e-5 if s == 0 else 1e-3, err_msg=f"step {s} vs oracle")
    np.testing.assert_allclose(losses[0], g["simclr_r18_losses"][0], rtol=1e-5)        # vs the reference's own numbers
    np.testing.assert_allclose(losses[1], g["simclr_r18_losses"][1], rtol=1e-3)
    np.testing.assert_allclose(losses[2], g["simclr_r18_losses"][2], rtol=2e-2)


def test_simclr_r50_step0_matches_reference(dev, golden):
    """ResNet-50 standard stem (7x7/2), 64x64, bs 8: step 0 is a pure function of the inputs."""
    g = golden["step_level"]
    m = _Step(dev, "resnet50", False)
    a1, a2 = seeded_randn(200, 8, 3, 64, 64), seeded_randn(201, 8, 3, 64, 64)
    loss, z1, z2 = m.step(a1, a2)
    np.testing.assert_allclose(loss, g["simclr_r50_losses"][0], rtol=2e-5)
    # bs 8 at 64x64 leaves 32 samples per layer4 BN channel and 8 per head BN column: ill-conditioned - the CPU
    # oracle itself is 9e-4 away from an fp64 evaluation of z here (tools/diag_grads.py resnet50 0 8 64)
    close(z1, g["simclr_r50_z1"], rtol=1e-4, atol=2e-3, what="z_1")
    close(z2, g["simclr_r50_z2"], rtol=1e-4, atol=2e-3, what="z_2")
    o = oracle.SimCLROracle("resnet50", False, 128, lr=float(g["simclr_r50_lr"]), weight_decay=1e-4)
    o.train_step(a1, a2)
    _check_grads(m, o, flip_tol=5e-2)


def _oracle64_like(o32_factory):
    """An fp64 copy of the oracle with the SAME (fp32-drawn) initial weights."""
    torch.set_default_dtype(torch.float64)
    try:
        o64 = o32_factory()
    finally:
        torch.set_default_dtype(torch.float32)
    src = o32_factory()
    for dst, s_ in ((o64.encoder, src.encoder), (o64.proj_head, src.proj_head)):
        for k in dst:
            if dst[k].dtype.is_floating_point:
                dst[k].data = s_[k].detach().double()
    return o64


def test_gradients_as_close_to_fp64_truth_as_the_cpu_path(dev):
    """Gradient parity, calibrated: ReLU mask flips make ANY two fp32 evaluations of this network differ by
    ~1e-3 in the early layers' gradients (see _check_grads), so the meaningful statement is about distance
    to the fp64 truth: per tensor, the HIP step must be as close to an fp64 oracle as the fp32 CPU oracle is
    (median within 3x, worst bounded by the flip size)."""
    a1, a2 = seeded_randn(300, 64, 3, 32, 32), seeded_randn(301, 64, 3, 32, 32)
    make = lambda: oracle.SimCLROracle("resnet18", True, 128, lr=0.2, weight_decay=1e-4)
    m, o32, o64 = _Step(dev, "resnet18", True), make(), _oracle64_like(make)
    loss, z1, _ = m.step(a1, a2)
    r32 = o32.train_step(a1, a2, return_z=True)
    r64 = o64.train_step(a1.double(), a2.double(), return_z=True)
    assert abs(loss - r64["loss"]) <= 3 * abs(r32["loss"] - r64["loss"]) + 2e-6 * abs(r64["loss"])
    ez_hip, ez_cpu = float((z1.cpu().double() - r64["z_1"]).abs().max()), float((r32["z_1"].double() - r64["z_1"]).abs().max())
    assert ez_hip <= 3 * ez_cpu + 1e-5 and ez_hip < 1e-4, (ez_hip, ez_cpu)            # north-star: |dz| <= 1e-4
    e_hip, e_cpu = [], []
    for p, g32, g64, off in zip(m.params(), o32.last_grads, o64.last_grads, m.optim.arena.offsets):
        if float(g64.norm()) < 1e-5:
            continue
        got = m.grads[off:off + p.numel()]
        got = got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape)
        e_hip.append(rel_l2(got, g64))
        e_cpu.append(rel_l2(g32, g64))
    assert np.median(e_hip) <= 3 * np.median(e_cpu) + 1e-5, (np.median(e_hip), np.median(e_cpu))
    assert max(e_hip) < 2e-2, max(e_hip)
    # the deepest layers see no flip downstream of them: there the comparison is tight in absolute terms
    assert min(e_hip) < 5e-5, min(e_hip)


def test_features_match_reference(dev, golden):
    from ssv_amd import ops
    g = golden["step_level"]
    m = _Step(dev, "resnet18", True)
    with torch.no_grad():
        z = m.proj_head(m.encoder(seeded_randn(400, 16, 3, 32, 32).to(dev)))
        f = ops.l2norm_fwd(z.contiguous(), True)[0]
    close(f, g["features_r18"], rtol=1e-4, atol=1e-5, what="build_features")


def test_two_stream_views_bitwise_equal_sequential(dev):
    """The two views on two HIP streams (separate scratch, separate gradient slabs, BN running-stat order kept by
    events) give the same bits as the sequential pass: loss, embeddings, every parameter and BN buffer after 2 steps."""
    res = []
    for dual in (False, True):
        m = _Step(dev, "resnet18", True)
        out = []
        for s in range(2):
            loss, z1, z2 = m.step(seeded_randn(900 + 2 * s, 48, 3, 32, 32), seeded_randn(901 + 2 * s, 48, 3, 32, 32), dual=dual)
            out.append((loss, z1.cpu(), z2.cpu()))
        torch.cuda.synchronize()
        res.append((out, m.optim.arena.data.cpu().clone(), {k: v.cpu().clone() for k, v in m.state().items() if "running" in k or "num_batches" in k}))
    (o0, p0, b0), (o1, p1, b1) = res
    for (l0, a0, c0), (l1, a1, c1) in zip(o0, o1):
        assert l0 == l1 and torch.equal(a0, a1) and torch.equal(c0, c1)
    assert torch.equal(p0, p1)
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


def test_train_step_is_bitwise_repeatable(dev):
    """Fixed-order reductions everywhere (split-K slabs, BN partials): two runs give identical bits."""
    outs = []
    for _ in range(2):
        m = _Step(dev, "resnet18", True)
        loss, z1, _ = m.step(seeded_randn(1, 32, 3, 32, 32), seeded_randn(2, 32, 3, 32, 32))
        outs.append((loss, z1.cpu(), m.optim.arena.data.cpu().clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def _bare_trainer(cls, dev, config, loader_len=1):
    """Build a trainer the way its __init__ does, minus dataloaders / output dirs / wandb."""
    from ssv_amd.utils import train_utils
    t = object.__new__(cls)
    t.config, t.device, t.train_loader = config, dev, [None] * loader_len
    torch.manual_seed(420)
    t._build("resnet18")
    t.scheduler, t.warmup_epochs = train_utils.get_scheduler({**config["scheduler"], "epochs": config["epochs"]}, optimizer=t.optim)
    return t


def test_input_stream_and_early_loss_read_change_no_bit(dev):
    """The loader builds the next batch's views on a stream of its own (hnn.input_stream) and train_step reads the loss from a pinned-memory copy
    enqueued before the backward (hnn.early_item), so the host runs one step ahead of the GPU.  Four steps of the SimCLR trainer fed by the loader:
    every loss and every parameter afterwards equal the serial form's (views on the ambient stream, .item() after the update) bit for bit."""
    from ssv_amd import nn as hnn
    from ssv_amd.models.simclr import SimCLR
    from ssv_amd.utils import data_utils
    rng = np.random.default_rng(11)
    imgs, labels = rng.integers(0, 256, size=(96, 40, 40, 3), dtype=np.uint8), rng.integers(0, 10, size=96)
    norm = {"mean": [0.5, 0.5, 0.5], "std": [0.25, 0.25, 0.25]}
    aug = {"color_jitter": {"brightness": 0.4, "contrast": 0.4, "saturation": 0.4, "hue": 0.1, "apply_prob": 0.8}, "random_gray": {"p": 0.2},
           "random_resized_crop": {"size": [32, 32], "scale": [0.2, 1.0]}, "random_flip": None, "to_tensor": None, "normalize": norm}
    tfs = {"train": aug, "test": {"center_crop": {"size": [32, 32]}, "to_tensor": None, "normalize": norm}}
    cfg = {"epochs": 100, "proj_dim": 128, "encoder": {"reduce_bottom_conv": True}, "optimizer": {"name": "sgd", "lr": 0.05, "weight_decay": 1e-6},
           "scheduler": {"name": "cosine", "warmup_epochs": 0}, "loss_fn": {"normalize": True, "temperature": 0.5}}
    runs = []
    for ahead in (True, False):
        prev = hnn._INPUT_STREAM, hnn._EARLY_LOSS
        hnn._INPUT_STREAM = hnn._EARLY_LOSS = ahead
        try:
            t = _bare_trainer(SimCLR, dev, cfg)
            loader = data_utils.GpuTwoViewLoader(imgs, labels, tfs, batch_size=24, shuffle=True, device=dev)
            losses = [t.train_step(batch)["loss"] for batch in loader]
            torch.cuda.synchronize()
            runs.append((losses, t.optim.arena.data.cpu().clone()))
        finally:
            hnn._INPUT_STREAM, hnn._EARLY_LOSS = prev
    (l0, p0), (l1, p1) = runs
    assert len(l0) == 4 and l0 == l1, (l0, l1)
    assert torch.equal(p0, p1)


def test_input_stream_is_ordered_behind_the_latest_producer_of_resident_data(dev):
    """A second producer in the same process (another loader, a dataset generated on the device): data enqueued on the ambient stream AFTER the side
    stream exists must still be complete when an input_stream block reads it - every entry waits for the latest hnn.data_ready event, not only the
    first one for the ambient stream.  And every spelling of a device shares one stream pair."""
    from ssv_amd import nn as hnn
    with hnn.input_stream(dev):
        pass                                                            # the side stream exists from here on (its one-time wait is spent)
    src = torch.empty(64 << 20, device=dev)
    for i in range(24):                                                 # ~6 GB of fills queued on the ambient stream: still running when the block below is enqueued
        src.fill_(float(i))
    hnn.data_ready(dev)
    with hnn.input_stream(dev) as ins:
        got = src[::4099].clone()
        ins.publish(got)
    torch.cuda.synchronize()
    assert bool((got == 23.0).all()), got[:8]
    assert hnn._view_stream_pair("cuda") is hnn._view_stream_pair(torch.device("cuda", torch.cuda.current_device())) is hnn._view_stream_pair(dev)
    assert hnn.input_stream("cuda").device == hnn.input_stream(dev).device


def test_barlow_r18_steps_match_reference(dev, golden):
    from ssv_amd.models.barlow import BarlowTwins
    g = golden["step_level"]
    cfg = {"epochs": 1000, "proj_dim": 256, "encoder": {"reduce_bottom_conv": True},
           "optimizer": {"name": "sgd", "lr": 0.2, "weight_decay": 1.5e-6}, "scheduler": {"name": "cosine", "warmup_epochs": 10},
           "loss_fn": {"normalize": False, "off_diagonal_weight": 0.005}}
    t = _bare_trainer(BarlowTwins, dev, cfg)
    assert abs(t.optim.param_groups[0]["lr"] - float(g["barlow_r18_lr"])) < 1e-15
    losses = [t.train_step({"aug_1": seeded_randn(300 + 2 * s, 32, 3, 32, 32), "aug_2": seeded_randn(301 + 2 * s, 32, 3, 32, 32)})["loss"]
              for s in range(2)]
    np.testing.assert_allclose(losses[0], g["barlow_r18_losses"][0], rtol=2e-5)
    np.testing.assert_allclose(losses[1], g["barlow_r18_losses"][1], rtol=2e-2)      # one update later (chaos: see r18 SimCLR test; Barlow's loss is ~300 x its gradient scale)


def test_byol_r18_steps_match_reference(dev, golden):
    from ssv_amd.models.byol import BYOL
    g = golden["step_level"]
    cfg = {"epochs": 1000, "proj_dim": 128, "tau": 0.996, "encoder": {"reduce_bottom_conv": True},
           "optimizer": {"name": "sgd", "lr": 0.2, "weight_decay": 1e-4}, "scheduler": {"name": "cosine", "warmup_epochs": 10}}
    t = _bare_trainer(BYOL, dev, cfg)                     # max_steps = 1000 * 1
    assert t.max_steps == 1000 and abs(t.optim.param_groups[0]["lr"] - float(g["byol_r18_lr"])) < 1e-15
    losses, taus = [], []
    for s in range(2):
        losses.append(t.train_step({"aug_1": seeded_randn(500 + 2 * s, 16, 3, 32, 32), "aug_2": seeded_randn(501 + 2 * s, 16, 3, 32, 32)})["loss"])
        t._after_step(s)                                  # update_tau(step) + momentum_update(), like the train loop
        taus.append(t.tau)
    np.testing.assert_allclose(losses[0], g["byol_r18_losses"][0], rtol=1e-5)
    np.testing.assert_allclose(losses[1], g["byol_r18_losses"][1], rtol=1e-2)      # one update later: size class only (chaos, see the SimCLR step test)
    np.testing.assert_allclose(taus, g["byol_r18_taus"], rtol=1e-12)
    state = {**{"online_network." + k: v for k, v in t.online_network.state_dict().items()},
             **{"target_network." + k: v for k, v in t.target_network.state_dict().items()}}
    for k, ref in zip(g["byol_r18_after2_keys"], g["byol_r18_after2_sums"]):
        k = str(k)
        got = oracle.tensor_checksum(state[k].contiguous())
        if k.startswith("target_network") and not ("running" in k):
            # target = EMA(tau ~ 0.996) of an independently initialised net: 99.6 % of it is init -> tight
            # (BN biases start at 0: what they hold is purely the EMA of the flip-noisy online gradient)
            np.testing.assert_allclose(got[1], ref[1], rtol=2e-2 if k.endswith("bias") else 1e-5, err_msg=k)
            np.testing.assert_allclose(got[0], ref[0], rtol=1e-4, atol=(2e-2 if k.endswith("bias") else 1e-3) * (float(np.sqrt(ref[1])) + 1e-6), err_msg=k)
        elif "running" in k:
            np.testing.assert_allclose(got[1], ref[1], rtol=2e-3, err_msg=k)


def test_bench_shape_r50_224_step0_matches_oracle(dev):
    """The BENCH workload's own shape (ResNet-50, 7x7/2 stem, 224x224) at a small ragged batch: step-0 loss within
    1e-4 relative and projected features within the CPU path's own fp64 distance (north-star bars)."""
    b = 6
    a1, a2 = seeded_randn(1000, b, 3, 224, 224), seeded_randn(1001, b, 3, 224, 224)
    m = _Step(dev, "resnet50", False)
    make = lambda: oracle.SimCLROracle("resnet50", False, 128, lr=0.2, weight_decay=1e-4)
    o32, o64 = make(), _oracle64_like(make)
    loss, z1, z2 = m.step(a1, a2, dual=True)
    r32 = o32.train_step(a1, a2, return_z=True)
    r64 = o64.train_step(a1.double(), a2.double(), return_z=True)
    np.testing.assert_allclose(loss, r32["loss"], rtol=1e-4)
    e_hip = float((z1.cpu().double() - r64["z_1"]).abs().max())
    e_cpu = float((r32["z_1"].double() - r64["z_1"]).abs().max())
    assert e_hip <= 3 * e_cpu + 1e-5, (e_hip, e_cpu)      # batch 6 feeds 6-sample BatchNorm1d columns: ill-conditioned for both
    _check_grads(m, o32, flip_tol=1e-1)


def test_ragged_last_batch(dev):
    """50000 mod 512 = 336 in the reference's loader (last batch is not dropped): a batch that is no multiple of
    any tile size must work and match."""
    b = 21
    a1, a2 = seeded_randn(1100, b, 3, 32, 32), seeded_randn(1101, b, 3, 32, 32)
    m = _Step(dev, "resnet18", True)
    o = oracle.SimCLROracle("resnet18", True, 128, lr=0.2, weight_decay=1e-4)
    loss, z1, _ = m.step(a1, a2)
    ref = o.train_step(a1, a2, return_z=True)
    np.testing.assert_allclose(loss, ref["loss"], rtol=1e-5)
    close(z1, ref["z_1"], rtol=1e-4, atol=1e-4, what="z_1")


def _arch_parity(dev, arch, factory, keys, sums, feats, xseed, dseed, batch):
    """One forward + backward of an encoder of main.py's --arch list on the HIP path: init draws (RNG-stream parity) and features against the
    reference's own numbers, features and every parameter gradient calibrated against an fp64 evaluation of the oracle."""
    from ssv_amd.networks import resnet
    torch.manual_seed(420)
    net = getattr(resnet, factory)(reduce_bottom_conv=True)
    sd = net.state_dict()
    assert [k for k, v in sd.items() if v.dtype.is_floating_point] == [str(k) for k in keys]
    for k, ref in zip(keys, sums):
        np.testing.assert_allclose(np.array(oracle.tensor_checksum(sd[str(k)].contiguous())), ref, rtol=1e-12, atol=0, err_msg=str(k))
    net = net.to(dev)
    x, dy = seeded_randn(xseed, batch, 3, 32, 32), seeded_randn(dseed, batch, 2048)
    y = net(x.to(dev))
    y.backward(dy.to(dev))

    def cpu(dtype):
        torch.manual_seed(420)
        p = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in oracle.init_resnet(arch, True).items()}
        for k, v in p.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        out = oracle.resnet_forward(p, x.to(dtype), arch, True)
        out.backward(dy.to(dtype))
        return p, out.detach()
    p64, y64 = cpu(torch.float64)
    p32, y32 = cpu(torch.float32)
    # features: 2x2 / 1x1 maps at this input size put only 4 * batch values into a channel's BatchNorm statistics, which amplifies rounding for any
    # fp32 evaluation - measured here as the fp32 CPU oracle's own distance to fp64 (e_cpu, 1e-5 .. 2e-4); HIP is held to that size class, and to
    # the reference's numbers with the same yardstick (the ResNet-18/50 step tests use 1e-4: their e_cpu is 2e-5)
    e_cpu = float((y32.double() - y64).abs().max())
    e_hip = float((y.detach().cpu().double() - y64).abs().max())
    assert e_hip <= 3 * e_cpu + 1e-5, (arch, e_hip, e_cpu)
    np.testing.assert_allclose(y.detach().cpu().numpy(), feats, rtol=1e-4, atol=max(1e-4, 4 * e_cpu))
    errs = {}
    for name, p in net.named_parameters():
        ref = p64[name].grad
        errs[name] = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30))
    # a ReLU whose input is within rounding of zero flips between any two fp32 evaluations and moves the gradients of everything
    # before it; with 16-32 values per channel in layer4 this is large here - calibrate against what the
    # fp32 CPU oracle itself shows against the same fp64 evaluation
    cpu_err = np.array([float((p32[k].grad.double() - p64[k].grad).norm() / (p64[k].grad.norm() + 1e-30)) for k in errs])
    vals = np.array(list(errs.values()))
    assert np.median(vals) <= 3 * np.median(cpu_err) + 1e-4 and vals.max() <= 3 * cpu_err.max() + 1e-3, \
        (arch, float(np.median(vals)), float(np.median(cpu_err)), float(vals.max()), float(cpu_err.max()))
    return net


def test_resnext50_grouped_convs_match_reference(dev, golden):
    """`-m resnext50`: grouped 3x3 convolutions run on group-aware tiles of the dense block-diagonal bank - init draws, features and every
    parameter gradient against the reference's own numbers (tests/golden/resnext_level.npz) and an fp64 evaluation of the oracle."""
    g = golden["resnext_level"]
    net = _arch_parity(dev, "resnext50", "resnext50_32x4d", g["init_keys"], g["init_sums"], g["features"], 1400, 1401, 4)
    grouped = net.layer1[0].conv2
    assert grouped.groups == 32 and tuple(grouped.weight.shape) == (128, 4, 3, 3)


@pytest.mark.parametrize("arch,factory", [("wide_resnet50", "wide_resnet50_2"), ("wide_resnet101", "wide_resnet101_2"), ("resnext101", "resnext101_32x8d")])
def test_remaining_archs_match_reference(dev, golden, arch, factory):
    """The rest of main.py's --arch list (`-m wide_resnet50 | wide_resnet101 | resnext101`, networks/resnet.py:174-193): 128 .. 1024-wide 3x3 layers
    (Winograd and fused-BatchNorm dispatch at widths ResNet-50 never shows) and 32 x 8d grouped layers, against tests/golden/arch_level.npz."""
    g = golden["arch_level"]
    net = _arch_parity(dev, arch, factory, g[f"{arch}_init_keys"], g[f"{arch}_init_sums"], g[f"{arch}_features"], 1700, 1701, 8)
    conv2 = net.layer1[0].conv2
    if arch == "resnext101":
        assert conv2.groups == 32 and tuple(conv2.weight.shape) == (256, 8, 3, 3)
    else:
        assert conv2.groups == 1 and tuple(conv2.weight.shape) == (128, 128, 3, 3)


@pytest.mark.parametrize("arch,rbc,size,b", [("resnet50", False, 64, 6), ("resnet18", True, 32, 20)])
def test_fused_batchnorm_chain_is_bitwise_the_materialised_one(dev, arch, rbc, size, b):
    """conv -> BN -> ReLU -> conv chains do not write the activation between the convolutions (the consumer applies scale / shift /
    ReLU while it stages its input, forward and weight gradient; the projection shortcut's BatchNorm is folded into the unit's
    closing kernel; the backward recomputes the ReLU gate from the forward's own scale / shift).  Same fmaf / fmaxf on the same
    floats as the stand-alone apply kernel: loss, embeddings, every gradient and the updated parameters must be IDENTICAL bits."""
    from ssv_amd import nn as hnn
    a1, a2 = seeded_randn(1500, b, 3, size, size), seeded_randn(1501, b, 3, size, size)
    outs = []
    # the projection shortcut's backward sums come from the gate epilogue only when that BatchNorm is folded (i.e. in the fused run): a
    # different summation order, covered by its own test - here both runs reduce it in its own pass so that the comparison stays bit for bit
    prev_gate, hnn._FUSE_SHORTCUT_GATE = hnn._FUSE_SHORTCUT_GATE, False
    # likewise the stem: its fused BatchNorm + ReLU + MaxPool (fused run only) reduces over the pooled positions by default - the same sums grouped per
    # window (test_fused_stem_batchnorm_relu_maxpool_against_the_three_kernel_form); here it walks the full-resolution map like the three-kernel form
    prev_pool, hnn._POOLED_STEM_REDUCE = hnn._POOLED_STEM_REDUCE, False
    try:
        for fuse in (True, False):
            prev, hnn._FUSE_BN_APPLY = hnn._FUSE_BN_APPLY, fuse
            try:
                m = _Step(dev, arch, rbc)
                loss, z1, z2 = m.step(a1, a2)
                torch.cuda.synchronize()
                outs.append((loss, z1.cpu(), z2.cpu(), m.grads.cpu().clone(), m.optim.arena.data.cpu().clone(),
                             {k: v.cpu().clone() for k, v in m.state().items() if "running" in k}))
            finally:
                hnn._FUSE_BN_APPLY = prev
    finally:
        hnn._FUSE_SHORTCUT_GATE = prev_gate
        hnn._POOLED_STEM_REDUCE = prev_pool
    f, u = outs
    assert f[0] == u[0] and torch.equal(f[1], u[1]) and torch.equal(f[2], u[2])
    assert torch.equal(f[3], u[3]), f"gradients differ: max {float((f[3] - u[3]).abs().max()):.3e}"
    assert torch.equal(f[4], u[4])
    for k in f[5]:
        assert torch.equal(f[5][k], u[5][k]), k


@pytest.mark.parametrize("arch,rbc,size,b", [("resnet50", False, 64, 6), ("resnet18", True, 32, 20), ("resnet50", False, 96, 3)])
def test_batchnorm_backward_reduced_in_the_conv_epilogue_matches_the_two_pass_form(dev, arch, rbc, size, b):
    """The data gradient that feeds a BatchNorm + ReLU backward gates itself and reduces (sum g, sum g * xhat) in its own epilogue
    (stride-1 layers on the forward kernel, stride-2 layers on the dgrad kernel, byte-mask and recomputed-gate variants); the
    stand-alone reduction pass is skipped.  Same ReLU bits, same sums in a different order: the forward is bit-identical and every
    gradient tensor agrees to rounding (1e-5 relative l2; no ReLU flip can occur, the masks are the forward's)."""
    from ssv_amd import nn as hnn
    a1, a2 = seeded_randn(1600, b, 3, size, size), seeded_randn(1601, b, 3, size, size)
    outs = []
    for fuse in (True, False):
        prev, hnn._FUSE_BN_BWD = hnn._FUSE_BN_BWD, fuse
        try:
            m = _Step(dev, arch, rbc)
            loss, z1, z2 = m.step(a1, a2)
            torch.cuda.synchronize()
            outs.append((loss, z1.cpu(), m, m.grads.cpu().clone()))
        finally:
            hnn._FUSE_BN_BWD = prev
    (lf, zf, m, gf), (lu, zu, _, gu) = outs
    assert lf == lu and torch.equal(zf, zu)
    worst = 0.0
    for p, off in zip(m.params(), m.optim.arena.offsets):
        a, r = gf[off:off + p.numel()].double(), gu[off:off + p.numel()].double()
        if float(r.norm()) < 1e-5:
            assert float(a.abs().max()) < 1e-5
            continue
        worst = max(worst, float((a - r).norm() / r.norm()))
    assert worst < 1e-5, f"worst per-tensor gradient difference {worst:.2e}"


@pytest.mark.parametrize("pooled", [False, True])
@pytest.mark.parametrize("arch,rbc,size,b", [("resnet50", False, 64, 5), ("resnet18", True, 32, 12), ("resnet50", False, 70, 3)])
def test_fused_stem_batchnorm_relu_maxpool_against_the_three_kernel_form(dev, arch, rbc, size, b, pooled):
    """The image stem's maxpool(relu(bn1(conv1(x)))) as ONE pass forward (pooled map + argmax slots out of the raw conv output) and
    one reduce + one apply pass backward (the full-resolution activation and its gradient are never written), odd sizes (35 x 35 feature
    map, ragged windows) included.  With the reduction walking the full-resolution map (pooled = False, SSV_NO_POOLED_STEM_REDUCE=1): identical
    bits to BatchNorm -> MaxPool -> their two backward kernels.  With the shipped reduction over the POOLED positions (dpool and the conv output
    kept at every window's arg-max): the forward is identical, the stem's gradients are the same sums grouped per window instead of per pixel -
    equal to rounding."""
    from ssv_amd import nn as hnn
    a1, a2 = seeded_randn(1700, b, 3, size, size), seeded_randn(1701, b, 3, size, size)
    outs = []
    prev_p, hnn._POOLED_STEM_REDUCE = hnn._POOLED_STEM_REDUCE, pooled
    try:
        for fuse in (True, False):
            prev, hnn._FUSE_STEM_POOL = hnn._FUSE_STEM_POOL, fuse
            try:
                m = _Step(dev, arch, rbc)
                loss, z1, z2 = m.step(a1, a2)
                torch.cuda.synchronize()
                outs.append((loss, z1.cpu(), z2.cpu(), m.grads.cpu().clone(), m.optim.arena.data.cpu().clone(), m))
            finally:
                hnn._FUSE_STEM_POOL = prev
    finally:
        hnn._POOLED_STEM_REDUCE = prev_p
    f, u = outs
    assert f[0] == u[0] and torch.equal(f[1], u[1]) and torch.equal(f[2], u[2])
    if not pooled:
        assert torch.equal(f[3], u[3]), f"gradients differ: max {float((f[3] - u[3]).abs().max()):.3e}"
        assert torch.equal(f[4], u[4])
        return
    m = f[5]
    worst, moved = 0.0, False
    for p, off in zip(m.params(), m.optim.arena.offsets):
        a, r = f[3][off:off + p.numel()].double(), u[3][off:off + p.numel()].double()
        if float(r.norm()) < 1e-5:
            assert float(a.abs().max()) < 1e-5
            continue
        e = float((a - r).norm() / r.norm())
        moved = moved or e > 0
        worst = max(worst, e)
    # only the stem's own tensors (conv1.weight, bn1.weight, bn1.bias) see the regrouped sums; everything behind the pooling layer is untouched
    assert worst < 1e-5, f"worst per-tensor gradient difference {worst:.2e}"
    assert moved or size < 40, "the pooled reduction did not run"


@pytest.mark.parametrize("arch,size,b", [("resnet50", 64, 6), ("resnet50", 96, 3)])
def test_batchnorm_backward_formed_by_its_consumers_matches_the_materialised_form(dev, arch, size, b):
    """ops.LazyGrad through a whole ResNet-50: behind conv3 and the stride-1 projection shortcut the BatchNorm backward hands (g, x,
    coefficients) to the convolution's weight / data gradient instead of writing dx.  Same inputs, same weights: the forward is
    bit-identical and every parameter gradient agrees with the materialised form to rounding level - the two differ only in how
    dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)) is associated (fused multiply-adds on the staged operand)."""
    from ssv_amd import nn as hnn
    a1, a2 = seeded_randn(1800, b, 3, size, size), seeded_randn(1801, b, 3, size, size)
    outs = []
    prev_hw, hnn._BN_DY_MIN_HW = hnn._BN_DY_MIN_HW, 0            # small inputs: every eligible layer, not only the >= 28x28 maps
    try:
        for fuse in (True, False):
            prev, hnn._FUSE_BN_DY = hnn._FUSE_BN_DY, fuse
            try:
                m = _Step(dev, arch, False)
                loss, z1, z2 = m.step(a1, a2)
                torch.cuda.synchronize()
                outs.append((loss, z1.cpu(), m, m.grads.cpu().clone()))
            finally:
                hnn._FUSE_BN_DY = prev
    finally:
        hnn._BN_DY_MIN_HW = prev_hw
    (lf, zf, m, gf), (lu, zu, _, gu) = outs
    assert lf == lu and torch.equal(zf, zu)
    worst = 0.0
    for p, off in zip(m.params(), m.optim.arena.offsets):
        a, r = gf[off:off + p.numel()].double(), gu[off:off + p.numel()].double()
        if float(r.norm()) < 1e-5:
            assert float(a.abs().max()) < 1e-5
            continue
        worst = max(worst, float((a - r).norm() / r.norm()))
    assert 0.0 < worst < 2e-5, f"worst per-tensor gradient difference {worst:.2e} (0 = the fused path did not run)"


@pytest.mark.parametrize("arch,rbc,size,b", [("resnet50", False, 64, 6), ("resnet50", False, 96, 3), ("resnet18", True, 32, 12)])
def test_closing_activation_formed_by_the_next_conv1_is_bitwise_the_element_wise_pass(dev, arch, rbc, size, b):
    """hnn.LazySum: the closing relu(bn3(x) + shortcut) of a bottleneck unit is formed AND written by conv1 of the next unit while it stages
    its input (identity and projection shortcuts, the stage-entry units whose conv1 now runs before the projection shortcut); ResNet-18's
    3x3 conv1 cannot, so its units fall back to the element-wise pass.  Same fmaf / add / fmaxf on the same floats: loss and embeddings are
    bit-identical; the gradients agree to rounding (at the stage entries the gate + partial sums of the unit-input gradient move from the
    stride-2 data-gradient kernel's epilogue to the forward kernel's: same ReLU bits, sums grouped differently)."""
    from ssv_amd import nn as hnn, ops
    a1, a2 = seeded_randn(1900, b, 3, size, size), seeded_randn(1901, b, 3, size, size)
    outs, calls = [], []
    inner = ops.conv2d_fwd_sumin

    def counted(*a, **k):
        calls[-1] += 1
        return inner(*a, **k)
    prev_hw, hnn._CLOSING_HW = hnn._CLOSING_HW, (0, 10 ** 9)     # small inputs: every stage, not only the >= 28x28 maps the shipped threshold selects
    ops.conv2d_fwd_sumin = counted
    try:
        for fuse in (True, False):
            prev, hnn._FUSE_CLOSING = hnn._FUSE_CLOSING, fuse
            calls.append(0)
            try:
                m = _Step(dev, arch, rbc)
                loss, z1, z2 = m.step(a1, a2)
                torch.cuda.synchronize()
                outs.append((loss, z1.cpu(), z2.cpu(), m, m.grads.cpu().clone()))
            finally:
                hnn._FUSE_CLOSING = prev
    finally:
        hnn._CLOSING_HW = prev_hw
        ops.conv2d_fwd_sumin = inner
    # ResNet-50: conv1 of 15 of its 16 units takes a deferred closing activation, per view; ResNet-18's 3x3 conv1 never can
    assert calls == ([30, 0] if arch == "resnet50" else [0, 0]), calls
    (lf, zf, z2f, m, gf), (lu, zu, z2u, _, gu) = outs
    assert lf == lu and torch.equal(zf, zu) and torch.equal(z2f, z2u)
    worst = 0.0
    for p, off in zip(m.params(), m.optim.arena.offsets):
        a, r = gf[off:off + p.numel()].double(), gu[off:off + p.numel()].double()
        if float(r.norm()) < 1e-5:
            assert float(a.abs().max()) < 1e-5
            continue
        worst = max(worst, float((a - r).norm() / r.norm()))
    assert worst < 1e-5, f"worst per-tensor gradient difference {worst:.2e}"       # (that the fused branch ran is what `calls` shows)


@pytest.mark.parametrize("arch,size,b", [("resnet50", 64, 6), ("resnet50", 96, 3)])
def test_projection_shortcut_backward_without_its_reduction_pass_matches_the_two_pass_form(dev, arch, size, b):
    """The gate epilogue that produces the gated gradient of a unit's closing activation also reduces it against the projection shortcut's
    BatchNorm input (ssv_bn_gate.x2), so that BatchNorm's backward skips its reduction pass (and, behind the stride-1 shortcut of the first
    stage, its element-wise pass too).  Forward untouched; every gradient tensor agrees with the two-pass form to rounding."""
    from ssv_amd import nn as hnn
    a1, a2 = seeded_randn(2000, b, 3, size, size), seeded_randn(2001, b, 3, size, size)
    outs = []
    prev_hw, hnn._BN_DY_MIN_HW = hnn._BN_DY_MIN_HW, 0
    try:
        for fuse in (True, False):
            prev, hnn._FUSE_SHORTCUT_GATE = hnn._FUSE_SHORTCUT_GATE, fuse
            try:
                m = _Step(dev, arch, False)
                loss, z1, z2 = m.step(a1, a2)
                torch.cuda.synchronize()
                outs.append((loss, z1.cpu(), m, m.grads.cpu().clone()))
            finally:
                hnn._FUSE_SHORTCUT_GATE = prev
    finally:
        hnn._BN_DY_MIN_HW = prev_hw
    (lf, zf, m, gf), (lu, zu, _, gu) = outs
    assert lf == lu and torch.equal(zf, zu)
    worst = 0.0
    for p, off in zip(m.params(), m.optim.arena.offsets):
        a, r = gf[off:off + p.numel()].double(), gu[off:off + p.numel()].double()
        if float(r.norm()) < 1e-5:
            assert float(a.abs().max()) < 1e-5
            continue
        worst = max(worst, float((a - r).norm() / r.norm()))
    assert 0.0 < worst < 2e-5, f"worst per-tensor gradient difference {worst:.2e} (0 = the fused path did not run)"


@pytest.mark.parametrize("arith,bar", [("f32", 2e-6), ("bf16x3", 1e-5)])
@pytest.mark.parametrize("size,b", [(64, 6), (96, 3), (70, 2)])
def test_compact_stride2_shortcut_gradient_matches_the_full_resolution_form(dev, size, b, arith, bar):
    """The data gradient of a stage entry's 1x1 / stride-2 projection shortcut (networks/resnet.py:131-135) stays compact - one dense GEMM on the
    subsampled grid (ops.StridedGrad) - and conv1's data gradient, the last contribution to the unit input's gradient, adds it in its gate
    epilogue at the pixels with even (h, w) (ssv_conv2d_fwd_gated_s2add / _dyin_s2add).  Against the full-resolution form (the parity-class
    kernel writing three quarters zeros, read back as a dense addend): forward untouched, every gradient equal to rounding; odd map sizes
    (35 -> 18, 9 -> 5) included.
    The two forms run the SAME product on two different kernels (forward kernel on the subsampled grid / parity-class data-gradient kernel).  On the fp32 MFMA
    instruction both are the same k-ordered fmaf chain and the gradients agree to 2e-6 - the bar of rounds 3-5, kept under SSV_ARITHMETIC=f32, where it proves that
    every pixel gets the right addend.  In the bf16x3 arithmetic the two kernels fold the piece products through differently shaped LDS images and tiles, i.e. two
    fp32-accurate evaluations with different rounding, and the difference - carried back through 50 layers - is bounded by 1e-5 (measured 2.8-3.7e-6), half the bar
    the fused / unfused BatchNorm chain above is held to."""
    from ssv_amd import nn as hnn, ops
    a1, a2 = seeded_randn(2100, b, 3, size, size), seeded_randn(2101, b, 3, size, size)
    outs, calls = [], []
    inner = ops.compact_s2_dgrad
    prev_hw, hnn._BN_DY_MIN_HW = hnn._BN_DY_MIN_HW, 0            # small inputs: the dy_in form of conv1's data gradient too
    try:
        with ops.arithmetic(arith):
            for compact in (True, False):
                prev, hnn._COMPACT_S2_DGRAD = hnn._COMPACT_S2_DGRAD, compact
                calls.append(0)

                def counted(*a, **k):
                    calls[-1] += 1
                    return inner(*a, **k)
                ops.compact_s2_dgrad = counted
                try:
                    m = _Step(dev, "resnet50", False)
                    loss, z1, z2 = m.step(a1, a2)
                    torch.cuda.synchronize()
                    outs.append((loss, z1.cpu(), m, m.grads.cpu().clone()))
                finally:
                    hnn._COMPACT_S2_DGRAD = prev
                    ops.compact_s2_dgrad = inner
    finally:
        hnn._BN_DY_MIN_HW = prev_hw
    assert calls == [6, 0], calls                                 # layer2 / layer3 / layer4 entries, two views
    (lf, zf, m, gf), (lu, zu, _, gu) = outs
    assert lf == lu and torch.equal(zf, zu)
    worst = 0.0
    for p, off in zip(m.params(), m.optim.arena.offsets):
        a, r = gf[off:off + p.numel()].double(), gu[off:off + p.numel()].double()
        if float(r.norm()) < 1e-5:
            assert float(a.abs().max()) < 1e-5
            continue
        worst = max(worst, float((a - r).norm() / r.norm()))
    assert worst < bar, f"worst per-tensor gradient difference {worst:.2e}"
