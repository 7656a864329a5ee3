"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol the
header declares, parameter init reproduces the reference's RNG stream and state_dict keys, and the
product refuses to run without a GPU (no silent fallback)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

import oracle
import ssv_amd
from ssv_amd import _lib
from ssv_amd.models import heads
from ssv_amd.networks import resnet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ssv_hip.h")).read()
    declared = set(re.findall(r"\b(ssv_[a-z0-9_]+)\s*\(", header))
    declared -= {"ssv_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libssv_hip.so lacks {name} declared in include/ssv_hip.h"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.ssv_version() >= 100


def test_init_matches_reference(golden):
    g = golden["init_checksums"]
    for tag, fn, kw, dim in (("r18rbc", resnet.resnet18, dict(reduce_bottom_conv=True), 512), ("r50", resnet.resnet50, {}, 2048)):
        torch.manual_seed(420)
        enc = fn(**kw)
        head = heads.SimclrProjectionHead(dim, 128)
        sd = enc.state_dict()
        assert list(sd.keys()) == list(g[f"{tag}_all_keys"])
        for keys, sums, d in ((g[f"{tag}_enc_keys"], g[f"{tag}_enc_sums"], sd), (g[f"{tag}_head_keys"], g[f"{tag}_head_sums"], head.state_dict())):
            for k, ref in zip(keys, sums):
                np.testing.assert_allclose(oracle.tensor_checksum(d[str(k)].contiguous()), ref, rtol=0, atol=0)
        n = sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in head.parameters())
        assert n == int(g[f"{tag}_nparams"])
        # conv filters are OHWI in memory
        assert enc.conv1.weight.is_contiguous(memory_format=torch.channels_last)


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    torch.manual_seed(0)
    enc = resnet.resnet18(reduce_bottom_conv=True)
    with pytest.raises(_lib.SsvError):
        enc(torch.randn(2, 3, 32, 32))
    from ssv_amd.utils import losses, train_utils
    with pytest.raises(_lib.SsvError):
        losses.SimclrLoss(True, 0.5)(torch.randn(4, 8), torch.randn(4, 8))
    with pytest.raises(_lib.SsvError):
        train_utils.get_optimizer({"name": "sgd", "lr": 0.1, "weight_decay": 0.0}, list(enc.parameters()))
    with pytest.raises(_lib.SsvError):
        enc.eval()


def test_lr_schedule_matches_reference(golden):
    """get_scheduler seeding + adjust_learning_rate, driven like models/simclr.py:77-84 (host-only logic)."""
    g = golden["optim_level"]
    from ssv_amd.utils import train_utils
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=2.0)          # any torch optimizer: only the scheduler logic is under test
    sched, warm = train_utils.get_scheduler({"name": "cosine", "warmup_epochs": 10, "epochs": 40}, optimizer=opt)
    rate = (2.0 - 1e-12) / warm
    lrs = [opt.param_groups[0]["lr"]]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for epoch in range(1, 31):
            if epoch <= warm:
                for grp in opt.param_groups:
                    grp["lr"] = 1e-12 + epoch * rate
            else:
                sched.step()
            lrs.append(opt.param_groups[0]["lr"])
    np.testing.assert_allclose(lrs, g["lr_schedule"], rtol=1e-12)


def _write_cifar10(root, n_train=50, n_test=20, seed=5):
    """A CIFAR-10 python archive in miniature: data_batch_1..5 + test_batch, b"data" [n, 3072] uint8 in CHW order, b"labels"."""
    import pickle
    base = os.path.join(root, "cifar-10-batches-py")
    os.makedirs(base)
    rng = np.random.default_rng(seed)
    files = {f"data_batch_{i}": n_train // 5 for i in range(1, 6)}
    files["test_batch"] = n_test
    truth = {}
    for name, n in files.items():
        data = rng.integers(0, 256, size=(n, 3072), dtype=np.uint8)
        labels = rng.integers(0, 10, size=n).tolist()
        with open(os.path.join(base, name), "wb") as fh:
            pickle.dump({b"data": data, b"labels": labels, b"batch_label": name.encode()}, fh)
        truth[name] = (data, labels)
    return truth


def test_cifar_pickles_are_read_into_nhwc_uint8(tmp_path):
    """the real-dataset path.  The python archive stores CHW rows; the GPU path wants [N,H,W,3] uint8 (reference
    utils/data_utils.py:8-11 hands torchvision.datasets.CIFAR10 the same files)."""
    from ssv_amd.utils import data_utils
    truth = _write_cifar10(str(tmp_path))
    x, y = data_utils._load_cifar(str(tmp_path), "cifar10", True)
    assert x.shape == (50, 32, 32, 3) and x.dtype == np.uint8 and x.flags["C_CONTIGUOUS"] and y.shape == (50,)
    want = np.concatenate([truth[f"data_batch_{i}"][0] for i in range(1, 6)]).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    np.testing.assert_array_equal(x, want)
    np.testing.assert_array_equal(y, np.concatenate([truth[f"data_batch_{i}"][1] for i in range(1, 6)]))
    xt, yt = data_utils._load_cifar(str(tmp_path), "cifar10", False)
    assert xt.shape == (20, 32, 32, 3) and yt.tolist() == truth["test_batch"][1]
    with pytest.raises(FileNotFoundError):
        data_utils._load_cifar(str(tmp_path / "nowhere"), "cifar10", True)
    with pytest.raises(AssertionError):
        data_utils._load("imagenet", str(tmp_path), None)


def test_batch_chunks_alignment_and_its_error():
    """ops._batch_chunks (host logic): chunks keep every tensor below the per-launch limit; with a statistics epilogue the chunk size
    is a multiple of 64 / gcd(rows per sample, 64) samples; when no such chunk fits, the error names that constraint (and
    conv2d_fwd_stats turns it into "no fused statistics" instead of failing)."""
    from ssv_amd import ops
    lim = ops._MAX_ELEMS
    assert ops._batch_chunks(4, (1000, 2000)) == [(0, 4)]
    per = lim // 10 + 1                                    # 9 samples fit one launch
    ch = ops._batch_chunks(40, (per, per // 2))
    assert ch[0] == (0, 9) and ch[-1][1] == 40 and all(b - a <= 9 for a, b in ch)
    ch = ops._batch_chunks(40, (per, per // 2), rows_per_sample=16)          # q = 4: chunks of 8 samples
    assert ch[0] == (0, 8) and all((b - a) % 4 == 0 for a, b in ch[:-1])
    with pytest.raises(_lib.SsvError, match="multiple of 64 samples"):
        ops._batch_chunks(40, (per, per // 2), rows_per_sample=49)           # odd rows per sample: q = 64 > 9
    with pytest.raises(_lib.SsvError, match="single sample"):
        ops._batch_chunks(2, (lim + 5, 10))


def test_dispatch_defaults_follow_the_arithmetic():
    """Round 6: three dispatch decisions trade multiplies (or VALU work) for bytes and were re-measured on the bf16x3 arithmetic with the traffic counters
    (profiles/r06_probe_wino_floor_traffic.txt, r06_probe_narrow_fuse.txt, r06_probe_thresholds_traffic.txt).  Their defaults follow ops.ARITHMETIC at CALL time -
    bench.py's fp32-instruction leg switches it inside one process - and an explicit setting (environment variable / module variable) overrides both."""
    from ssv_amd import nn as hnn
    from ssv_amd import ops
    keep = (ops.ARITHMETIC, ops.WINOGRAD44_MIN_CHANNELS, hnn._BN_DY_MIN_HW, hnn._CLOSING_HW)
    try:
        ops.WINOGRAD44_MIN_CHANNELS, hnn._BN_DY_MIN_HW, hnn._CLOSING_HW = None, None, None
        ops.ARITHMETIC = "bf16x3"
        assert (ops._wino44_min_channels(), hnn._bn_dy_min_hw(), hnn._closing_hw()) == (128, 196, (196, 10 ** 9))
        ops.ARITHMETIC = "f32"
        assert (ops._wino44_min_channels(), hnn._bn_dy_min_hw(), hnn._closing_hw()) == (64, 784, (784, 10 ** 9))
        ops.WINOGRAD44_MIN_CHANNELS, hnn._BN_DY_MIN_HW, hnn._CLOSING_HW = 256, 0, (0, 100)
        for a in ("bf16x3", "f32"):
            ops.ARITHMETIC = a
            assert (ops._wino44_min_channels(), hnn._bn_dy_min_hw(), hnn._closing_hw()) == (256, 0, (0, 100))
    finally:
        ops.ARITHMETIC, ops.WINOGRAD44_MIN_CHANNELS, hnn._BN_DY_MIN_HW, hnn._CLOSING_HW = keep


def test_dispatch_predicates_of_the_round3_paths():
    """Host logic that decides which kernel family a layer runs on (no compute): Winograd F(2x2, 3x3) only for 3x3 / stride 1 / padding 1 layers with
    >= 128 channels on both sides and enough tiles (networks/resnet.py:7-10,56-58: conv2 of ten ResNet-50 units), the row-taps stem only for a
    3-channel input with at most 8 taps per filter row (networks/resnet.py:96-99), the fused Linear backward only for 1x1 / stride-1 layers."""
    from ssv_amd import ops
    wino = lambda c, k, r, st, pad, shape: ops.use_winograd((k, c, r, r), st, pad, shape, False)
    assert wino(128, 128, 3, 1, 1, (512, 28, 28, 128)) and wino(256, 256, 3, 1, 1, (512, 14, 14, 256)) and wino(512, 512, 3, 1, 1, (512, 7, 7, 512))
    # layer1 (64 channels): F(2x2) measured slower (profiles/r03_probe_winograd.txt) - it runs Winograd only where ALL products take F(4x4) (round 5: enough
    # tiles, the weight gradient on F(4x4)), never F(2x2)
    # ... and (round 6) only on the fp32-MFMA arithmetic: on bf16x3 the direct product of those layers is faster and moves fewer bytes
    # (profiles/r06_probe_wino_floor_traffic.txt), unless SSV_WINOGRAD44_MIN_CHANNELS says otherwise
    keep_arith, keep_floor = ops.ARITHMETIC, ops.WINOGRAD44_MIN_CHANNELS
    try:
        ops.WINOGRAD44_MIN_CHANNELS = None
        ops.ARITHMETIC = "bf16x3"
        assert not wino(64, 64, 3, 1, 1, (512, 56, 56, 64)) and wino(128, 128, 3, 1, 1, (512, 28, 28, 128))
        ops.WINOGRAD44_MIN_CHANNELS = 64
        assert wino(64, 64, 3, 1, 1, (512, 56, 56, 64))
        ops.WINOGRAD44_MIN_CHANNELS = None
        ops.ARITHMETIC = "f32"
        assert wino(64, 64, 3, 1, 1, (512, 56, 56, 64)) and not wino(64, 64, 3, 1, 1, (4, 56, 56, 64)) and not wino(32, 32, 3, 1, 1, (512, 56, 56, 32))
        for name, off in (("WINOGRAD44", False), ("WINOGRAD44_WGRAD", False), ("WINOGRAD44_MIN_CHANNELS", 128)):
            keep = getattr(ops, name)
            setattr(ops, name, off)
            try:
                assert not wino(64, 64, 3, 1, 1, (512, 56, 56, 64)), name
            finally:
                setattr(ops, name, keep)
    finally:
        ops.ARITHMETIC, ops.WINOGRAD44_MIN_CHANNELS = keep_arith, keep_floor
    assert not wino(128, 128, 3, 2, 1, (512, 56, 56, 128))         # stride 2
    assert not wino(128, 128, 1, 1, 0, (512, 28, 28, 128))         # 1x1
    assert not wino(128, 128, 3, 1, 1, (2, 8, 8, 128))             # 32 tiles: not worth three extra launches
    assert not wino(144, 144, 3, 1, 1, (512, 28, 28, 144))         # a channel count the transforms' lane mapping does not take
    prev = ops.WINOGRAD
    ops.WINOGRAD = False
    try:
        assert not wino(256, 256, 3, 1, 1, (512, 14, 14, 256))
    finally:
        ops.WINOGRAD = prev
    assert ops.can_row_stem((64, 3, 7, 7)) and ops.can_row_stem((64, 3, 3, 3))
    assert not ops.can_row_stem((64, 4, 7, 7)) and not ops.can_row_stem((64, 3, 11, 11)) and not ops.can_row_stem((62, 3, 7, 7))
    assert ops.can_form_closing_sum((64, 256, 1, 1), 1, 0) and not ops.can_form_closing_sum((64, 256, 3, 3), 1, 1)
    assert not ops.can_form_closing_sum((64, 256, 1, 1), 1, 0, groups=32)


def test_dispatch_predicates_of_the_round4_paths():
    """Host logic, no compute: which product of a Winograd layer takes F(4x4, 3x3) (by the transformed-domain work it leaves, the contraction length its numerics bar
    was measured for, and the tile count - networks/resnet.py:7-10,56-58: conv2 of the 28x28 / 14x14 / 7x7 ResNet-50 units), and which feed-forward blocks take
    their GELU derivative in fc1's forward epilogue (networks/vit.py:46-60)."""
    from ssv_amd import ops
    fwd, dgr = ops.WINOGRAD44_MAX_RATIO_FWD, ops.WINOGRAD44_MAX_RATIO_DGRAD
    assert abs(ops._wino44_ratio(28, 28) - 0.5625) < 1e-12 and abs(ops._wino44_ratio(7, 7) - 0.5625) < 1e-12
    assert abs(ops._wino44_ratio(14, 14) - 36 * 16 / (16 * 49)) < 1e-12 and ops._wino44_ratio(2, 2) > 1
    prev = ops.WINOGRAD44
    ops.WINOGRAD44 = True
    try:
        assert ops._use_wino44(512, 28, 28, 128, 128, fwd) and ops._use_wino44(512, 28, 28, 128, 128, dgr)
        assert not ops._use_wino44(512, 14, 14, 256, 256, fwd) and ops._use_wino44(512, 14, 14, 256, 256, dgr)       # F(2x2) tiles 14x14 exactly: only the data gradient switches
        assert ops._use_wino44(512, 7, 7, 512, 512, fwd) and ops._use_wino44(512, 7, 7, 512, 512, dgr)
        assert not ops._use_wino44(512, 7, 7, 1024, 1024, dgr) and not ops._use_wino44(512, 7, 7, 512, 1024, fwd)    # wide_resnet's layer4: past the measured numerics bar
        assert not ops._use_wino44(64, 7, 7, 512, 512, fwd)                                                          # 256 tiles: 36 small GEMMs buy nothing over 16
        assert ops._use_wino44(256, 7, 7, 512, 512, fwd) and not ops._use_wino44(255, 7, 7, 512, 512, fwd)           # the boundary is ops.WINOGRAD44_MIN_TILES tiles
        ops.WINOGRAD44 = False
        assert not ops._use_wino44(512, 28, 28, 128, 128, dgr)
    finally:
        ops.WINOGRAD44 = prev
    pg, pl = ops.GELU_DACT_IN_FWD, ops.LINEAR_GELUGRAD_ON_FWD
    ops.GELU_DACT_IN_FWD = ops.LINEAR_GELUGRAD_ON_FWD = True
    try:
        assert ops.can_gelu_dact((3072, 768), (768, 3072)) and ops.can_gelu_dact((2048, 256), (256, 2048))           # ViT-B blocks, the DINO head's hidden layers
        assert not ops.can_gelu_dact((64, 768), (768, 64)) and not ops.can_gelu_dact((3072, 768), (10, 3072))        # too narrow / an output width the GEMM tile does not take
        ops.GELU_DACT_IN_FWD = False
        assert not ops.can_gelu_dact((3072, 768), (768, 3072))
    finally:
        ops.GELU_DACT_IN_FWD, ops.LINEAR_GELUGRAD_ON_FWD = pg, pl


def test_host_run_ahead_helpers_are_inert_without_a_gpu():
    """hnn.early_item / hnn.input_stream (the host runs one step ahead of the GPU) fall back to the plain forms on the CPU: .item(), ambient stream."""
    from ssv_amd import nn as hnn
    t = torch.tensor(3.25)
    e = hnn.early_item(t)
    assert e.ev is None and e.get() == 3.25
    with hnn.input_stream(torch.device("cpu")) as ins:
        assert not ins.enabled
        ins.publish(t, None, 5)
    assert hnn.input_stream("cpu").enabled is False


def test_profiling_aggregators_classify_every_kernel_like_the_scope_it_runs_under():
    """tools/kernel_classes.py (one table for pmc_traffic / pmc_mfma / kstats_steady) against the sources: every kernel a host function launches under a
    ProfScope classifies into that scope's class - bench.py divides a class's replayed counters by that class's HIP-event time (round 4: the stem's
    rows-in-LDS kernels were timed as conv_fwd / conv_wgrad and counted as 'other')."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_classes as kc
    rows = kc.launches_by_scope(os.path.join(ROOT, "self-supervised-vision_amd", "csrc"))
    assert len(rows) > 60 and sum(len(k) for _, _, _, k in rows) > 150                    # the parser still finds the launches
    bad = [(f, fn, k, kc.classify(k), sc) for f, fn, sc, ks in rows if fn not in kc.SCOPE_EXCEPTIONS for k in ks if kc.classify(k) not in sc]
    assert not bad, bad
    from ssv_amd import _lib
    assert {kc.classify(k) for _, _, _, ks in rows for k in ks} <= set(_lib.PROF_CLASSES)
    for name, cls in (("stem_fwd_rows_k<true>", "conv_fwd"), ("stem_wgrad_rows_k", "conv_wgrad"), ("wino44_dy_k", "conv_wgrad"), ("wgrad_reduce64_k", "conv_wgrad"),
                      ("void (anonymous namespace)::wino44_output_k<2>(int, int)", "conv_dgrad"), ("wino44_output_k<1>", "conv_fwd"), ("ntxent_merge_k", "loss")):
        assert kc.classify(name) == cls and kc.in_conv_family(name) == cls.startswith("conv_"), name


def test_step_graph_falls_back_to_the_eager_step_where_a_graph_cannot_be(monkeypatch):
    """graph.StepGraph's host logic without a GPU: the reasons it refuses (mode 0, unsafe trainer, wrong optimizer, CPU batch, large images under auto) all end in
    trainer.train_step(batch) being called - the step never disappears - and the graph's key carries every scalar a captured step bakes in (and no more)."""
    from ssv_amd import graph
    from ssv_amd.utils import train_utils

    class _Opt:
        param_groups = [{"lr": 0.2, "weight_decay": 1e-4, "momentum": 0.9, "nesterov": True}]
        _steps, clip = 3, 0.0

    class _T:
        graph_safe, graph_inputs = True, ("aug_1", "aug_2")
        optim = _Opt()
        calls = 0

        def train_step(self, batch):
            self.calls += 1
            return {"loss": 1.5}

        def graph_key(self):
            return (0.1, 0.04)

    batch = {"aug_1": torch.zeros(2, 3, 8, 8), "aug_2": torch.zeros(2, 3, 8, 8), "label": torch.zeros(2)}
    t = _T()
    sg = graph.StepGraph(t, mode="1")
    assert sg(batch) == {"loss": 1.5} and t.calls == 1 and "neither the fused SGD" in sg.describe()["disabled"]
    monkeypatch.setattr(train_utils, "FusedSGD", _Opt)                       # now the optimizer qualifies: the CPU batch is what stops it
    sg = graph.StepGraph(t, mode="1")
    assert sg(batch) == {"loss": 1.5} and "not on the GPU" in sg.describe()["disabled"]
    assert sg(batch) == {"loss": 1.5} and t.calls == 3                       # disabled: straight to the eager step
    t.graph_safe = False
    sg = graph.StepGraph(t, mode="1")
    sg(batch)
    assert "graph_safe" in sg.describe()["disabled"]
    t.graph_safe = True
    sg = graph.StepGraph(t, mode="0")
    sg(batch)
    assert sg.describe()["disabled"] == "SSV_STEP_GRAPH=0"
    key = graph.StepGraph(t, mode="1")._key({k: batch[k] for k in t.graph_inputs})
    assert key[1] == () and key[3] == (0.1, 0.04) and key[0][0][1] == (2, 3, 8, 8)          # SGD: lr / weight decay / momentum are device memory under capture, not baked in
    t.optim.param_groups[0]["lr"] = 0.1
    assert graph.StepGraph(t, mode="1")._key({k: batch[k] for k in t.graph_inputs}) == key     # the schedule moved four device floats: the same graph
    t.graph_key = lambda: (0.1, 0.05)
    assert graph.StepGraph(t, mode="1")._key({k: batch[k] for k in t.graph_inputs}) != key     # a scalar that still reaches a kernel as an argument moved: another graph


def test_step_graph_lifetime_rules():
    """graph.StepGraph: held weakly by the trainer's own step() (no cycle: the graphs die with the trainer, by reference count), strongly when built by hand; every
    live one is emptied by the interpreter-exit hook while the HIP runtime is still up."""
    import gc
    from ssv_amd import graph

    class T:
        graph_safe = True
    t = T()
    weak = graph.StepGraph(t, weak=True)
    strong = graph.StepGraph(T())
    assert weak.trainer is t and isinstance(strong.trainer, T)
    del t
    gc.collect()
    with pytest.raises(RuntimeError, match="no longer exists"):
        weak.trainer
    strong.graphs["k"] = ("graph", {}, None, {})
    graph._LIVE.add(strong)
    graph._close_all()
    assert strong.graphs == {}
