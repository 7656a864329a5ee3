"""GPU: gradient and multi-step parity AT THE BENCH NETWORK - ResNet-50 (7x7/2 stem), 224 x 224, batch 32, the default fusion thresholds
(every formed-on-load / gated / deferred variant the bs 512 step launches is live at this size).

What is and is not a well-posed statement here was measured first (tools/diag_r50_cpu_conditioning.py, tools/diag_r50_gpu.py):

  * Step 0 is a pure function of the inputs: loss to 6e-7, embeddings to 3e-4 (the fp32 CPU oracle: 3e-4 - this network amplifies a relative
    perturbation ~5000x on its way to the head).
  * Gradients: a ReLU of the head flips for a 1e-4 forward difference, so ANY two fp32 evaluations differ by ~2e-2 in every tensor upstream
    of it (fp32 CPU oracle vs its own fp64 twin: median 1.9e-2, worst 2.4e-2).  The HIP step is held to the CPU path's own distance to
    fp64, per tensor, plus a tight bound on the tensors that have no ReLU downstream.
  * Trajectories: the stem's gradient is 180x its weight norm (|g| 610 vs |w| 3.4: no normalisation on the way back through 16 units), so
    one update moves the stem by 35 % at lr 2e-3 and by 0.3 % at lr 1e-5 - and the fp32 CPU oracle is 1e-3 .. 5e-3 from its fp64 twin on
    step 1 at every learning rate from 1e-5 to 2e-3, always ~0.1 .. 0.3 of what the step moved the loss.  A learning rate at which two
    fp32 evaluations stay within 1e-5 while a step still moves the loss by more than the 1e-4 bar does not exist for this network and batch.
    Two statements replace it: (1) the free-running trajectory stays inside the fp64 envelope of the CPU path on every step; (2) with the
    state re-synchronised before every step ("teacher forcing": weights and momentum of the CPU trajectory at config lr / 100, where one
    update already turns the stem by a third - at the config's own 0.2 the first update of this batch-32 problem collapses the embeddings and
    every later loss is the constant ln(2B - 1)) each step is a pure function again - loss to 1e-5, the applied update as close to an fp64
    evaluation of the same step as the CPU path's.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import corr_views
from test_gpu_step import _Step, _oracle64_like, rel_l2

pytestmark = pytest.mark.gpu
BAR = 1e-4
B, SIZE = 32, 224


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _names(o):
    return [k for k in o.state() if k.endswith(".weight") or k.endswith(".bias")]


def _hip_grads(m):
    out = []
    for p, off in zip(m.params(), m.optim.arena.offsets):
        got = m.grads[off:off + p.numel()]
        out.append(got.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else got.view(p.shape))
    return out


def _default_thresholds():
    from ssv_amd import nn as hnn
    assert hnn._BN_DY_MIN_HW is None and hnn._CLOSING_HW is None and hnn._FUSE_BN_APPLY and hnn._FUSE_BN_BWD and hnn._FUSE_BN_DY \
        and hnn._FUSE_CLOSING and hnn._FUSE_SHORTCUT_GATE and hnn._FUSE_STEM_POOL, "this test pins the SHIPPED kernel selection"


def test_r50_224_step0_gradients_and_free_running_trajectory_inside_the_fp64_envelope(dev):
    _default_thresholds()
    m = _Step(dev, "resnet50", False, lr=0.02)                     # config lr / 100, seeded to 2e-3 (the bench gate's setting)
    lr = m.optim.param_groups[0]["lr"]
    make = lambda: oracle.SimCLROracle("resnet50", False, 128, lr=lr, weight_decay=1e-4)
    o32, o64 = make(), _oracle64_like(make)
    names = _names(o32)
    worst = 0.0
    for s in range(3):
        a1, a2 = corr_views(7000 + 3 * s, B, SIZE)
        loss, z1, _ = m.step(a1, a2, dual=True)                    # two view streams, as the timed step
        r32 = o32.train_step(a1, a2, return_z=True)
        r64 = o64.train_step(a1.double(), a2.double(), return_z=True)
        d_hip, d_cpu = abs(loss - r64["loss"]) / abs(r64["loss"]), abs(r32["loss"] - r64["loss"]) / abs(r64["loss"])
        worst = max(worst, d_cpu)
        if s == 0:
            assert abs(loss - r32["loss"]) <= 1e-5 * abs(r32["loss"]), (loss, r32["loss"])         # north-star: 1e-4
            e_hip = float((z1.cpu().double() - r64["z_1"]).abs().max())
            e_cpu = float((r32["z_1"].double() - r64["z_1"]).abs().max())
            assert e_hip <= 3 * e_cpu + 1e-5, (e_hip, e_cpu)       # measured 3.3e-4 vs 3.2e-4
            eh, ec, tight = [], [], []
            for got, g32, g64, name in zip(_hip_grads(m), o32.last_grads, o64.last_grads, names):
                if float(g64.norm()) < 1e-5:
                    assert float(got.abs().max()) < 1e-5, name
                    continue
                eh.append(rel_l2(got, g64))
                ec.append(rel_l2(g32, g64))
                if name in ("proj_head.fc2.weight", "proj_head.bn2.weight", "proj_head.bn2.bias", "proj_head.bn1.weight"):
                    tight.append((name, eh[-1]))                    # no ReLU mask between these tensors and the loss
            # measured: hip median 1.92e-2 / worst 2.72e-2, cpu32 median 1.92e-2 / worst 2.44e-2
            assert np.median(eh) <= 3 * np.median(ec) + 1e-5, (np.median(eh), np.median(ec))
            assert max(eh) <= 3 * max(ec), (max(eh), max(ec))
            assert len(tight) == 4 and all(e < 1e-3 for _, e in tight), tight                   # measured 7e-5 .. 1.2e-4
        # every step: no further from the fp64 trajectory than 3x the furthest the CPU path has been so far (+ the bar)
        assert d_hip <= 3 * worst + BAR, f"step {s}: hip {loss:.7f} cpu32 {r32['loss']:.7f} fp64 {r64['loss']:.7f}"


def _load_state(m, o, step):
    """Weights and momentum of the oracle -> the HIP trainer (test plumbing: torch copies into the arena views)."""
    from ssv_amd import ops
    from ssv_amd.utils.train_utils import ParamArena
    for p, off, src, buf in zip(m.params(), m.optim.arena.offsets, o.params, o.bufs):
        p.data.copy_(src.detach().to(p.device))
        mv = ParamArena._view(m.optim.momentum_buffer, p, off)
        if buf is None:
            mv.zero_()
        else:
            mv.copy_(buf.to(p.device))
    m.optim._steps = step
    ops.invalidate_weight_caches()


def test_r50_224_teacher_forced_steps(dev):
    import math
    _default_thresholds()
    m = _Step(dev, "resnet50", False, lr=0.02)                      # configs/simclr.yaml's lr / 100, seeded to 2e-3
    lr = m.optim.param_groups[0]["lr"]
    assert abs(lr - (1e-12 + 0.002)) < 1e-12
    seen = []
    make = lambda: oracle.SimCLROracle("resnet50", False, 128, lr=lr, weight_decay=1e-4)
    o32, o64 = make(), _oracle64_like(make)
    for s in range(2):                                              # step 1 starts from moved weights AND a filled momentum buffer
        a1, a2 = corr_views(7100 + 3 * s, B, SIZE)
        # every evaluation starts this step from the CPU trajectory's state
        with torch.no_grad():
            for dst, src in zip(o64.params, o32.params):
                dst.copy_(src.double())
            o64.bufs = [None if b is None else b.double().clone() for b in o32.bufs]
        _load_state(m, o32, s)
        before = [p.detach().clone() for p in o32.params]
        loss, _, _ = m.step(a1, a2, dual=bool(s & 1))
        r32 = o32.train_step(a1, a2)
        r64 = o64.train_step(a1.double(), a2.double())
        assert abs(loss - r32["loss"]) <= 1e-5 * abs(r32["loss"]), f"step {s}: hip {loss} cpu32 {r32['loss']} fp64 {r64['loss']}"
        assert abs(loss - math.log(2 * B - 1)) > 1e-2 and all(abs(loss - x) > 1e-4 for x in seen), f"degenerate trajectory: {seen + [loss]}"
        seen.append(loss)
        eh, ec = [], []
        for p, b0, p32, p64 in zip(m.params(), before, o32.params, o64.params):
            d64 = p64.detach() - b0.double()
            if float(d64.norm()) < 1e-9:
                continue
            eh.append(rel_l2(p.detach().cpu().double() - b0.double(), d64))
            ec.append(rel_l2(p32.detach().double() - b0.double(), d64))
        # the applied update (gradient + weight decay + Nesterov momentum of the carried buffer) against the fp64 evaluation of the same step
        assert np.median(eh) <= 3 * np.median(ec) + 1e-5 and max(eh) <= 3 * max(ec) + 1e-4, \
            f"step {s}: update error hip median {np.median(eh):.2e} worst {max(eh):.2e}, cpu32 median {np.median(ec):.2e} worst {max(ec):.2e}"
        print(f"step {s}: loss {loss:.6f} rel {abs(loss - r32['loss']) / abs(r32['loss']):.1e}; update error vs fp64 - hip median {np.median(eh):.2e} worst {max(eh):.2e}, "
              f"cpu32 median {np.median(ec):.2e} worst {max(ec):.2e}")


@pytest.mark.parametrize("size", [16, 28, 40])
def test_small_inputs_behind_the_7x7_stem_take_the_three_kernel_path(dev, size):
    """Inputs below 32 px leave the stem a map narrower than the fused BatchNorm+ReLU+MaxPool backward's row stride (RT = 16 pixels for 64
    channels): they must run (the reference accepts any size) and match the oracle - through the stand-alone BatchNorm and MaxPool kernels."""
    from conftest import seeded_randn
    b = 8
    a1, a2 = seeded_randn(7200, b, 3, size, size), seeded_randn(7201, b, 3, size, size)
    m = _Step(dev, "resnet50", False)
    o = oracle.SimCLROracle("resnet50", False, 128, lr=0.2, weight_decay=1e-4)
    loss, z1, _ = m.step(a1, a2)
    ref = o.train_step(a1, a2, return_z=True)
    np.testing.assert_allclose(loss, ref["loss"], rtol=1e-4)
    assert torch.isfinite(m.grads).all()
