"""GPU: Winograd F(2x2, 3x3) (csrc/winograd.hip + the batched implicit-GEMM launches) against torch fp64 convolutions - forward with the
fused input BatchNorm and the statistics epilogue, data gradient with the ReLU gate epilogue, weight gradient from the kept transformed
input - and the dispatch rule that sends the deep stride-1 3x3 layers of networks/resnet.py:56-58 through it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import seeded_randn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def f22_only():
    """This file pins F(2x2, 3x3) itself: ops.wino_conv2d_* must not hand over to F(4x4) (tests/test_gpu_winograd44.py) here."""
    from ssv_amd import ops
    prev, ops.WINOGRAD44 = ops.WINOGRAD44, False
    yield
    ops.WINOGRAD44 = prev


def rel(a, b):
    return float((a.detach().cpu().double() - b).norm() / (b.norm() + 1e-30))


SHAPES = [(6, 14, 14, 256, 256), (5, 7, 7, 512, 512), (3, 28, 28, 128, 128), (4, 8, 6, 128, 256), (3, 9, 5, 256, 128), (2, 7, 8, 1024, 128), (2, 4, 4, 128, 2048)]


def _case(dev, n, h, w_, c, k, seed=0):
    x = seeded_randn(seed + 1, n, h, w_, c)
    w = seeded_randn(seed + 2, k, c, 3, 3) * (2.0 / (9 * c)) ** 0.5
    dy = seeded_randn(seed + 3, n, h, w_, k)
    aff = (seeded_randn(seed + 4, c) * 0.3 + 1.0, seeded_randn(seed + 5, c) * 0.2)
    return x, w, dy, aff


@pytest.mark.parametrize("n,h,w_,c,k", SHAPES)
@pytest.mark.parametrize("affine", [False, True])
def test_winograd_forward_and_its_statistics_epilogue(dev, n, h, w_, c, k, affine):
    from ssv_amd import _lib, ops
    x, w, _, aff = _case(dev, n, h, w_, c, k)
    a = torch.relu(x.double() * aff[0].double() + aff[1].double()) if affine else x.double()
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    xd, wd = x.to(dev), w.contiguous(memory_format=torch.channels_last).to(dev)
    affd = (aff[0].to(dev), aff[1].to(dev)) if affine else None
    rpg = int(_lib.load().ssv_wino_stats_rows_per_group(n, h, w_))
    assert rpg == (64 if (h % 2 == 0 and w_ % 2 == 0) else (h * w_ if ((h + 1) // 2) * ((w_ + 1) // 2) == 16 else 0))
    y, part, v = ops.wino_conv2d_fwd(xd, wd, in_affine=affd, want_stats=rpg > 0, keep_v=True)
    assert rel(y, ref) < 2e-6, rel(y, ref)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5 * float(ref.abs().max()))
    assert tuple(v.shape) == (16, n * ((h + 1) // 2) * ((w_ + 1) // 2), c)
    if rpg:
        assert part[2] == rpg and part[0].shape == ((n * h * w_ + rpg - 1) // rpg, k)
        gamma, beta = torch.ones(k, device=dev), torch.zeros(k, device=dev)
        rm, rv, nbt = torch.zeros(k, device=dev), torch.ones(k, device=dev), torch.zeros((), dtype=torch.long, device=dev)
        mean, invstd, scale, shift = ops.bn_stats_finalize(n * h * w_, k, part, gamma, beta, rm, rv, nbt)
        r2 = ref.reshape(-1, k)
        np.testing.assert_allclose(mean.cpu().double().numpy(), r2.mean(0).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(invstd.cpu().double().numpy(), (1.0 / torch.sqrt(r2.var(0, unbiased=False) + 1e-5)).numpy(), rtol=1e-4)


@pytest.mark.parametrize("n,h,w_,c,k", SHAPES)
def test_winograd_data_gradient_with_the_relu_gate_epilogue(dev, n, h, w_, c, k):
    """dx = gate(conv_transpose(dy, w)): gate bit = (gx * scale + shift > 0) of the BatchNorm in front; the epilogue's partial sums add up to
    sum g and sum g * xhat over all rows (what ssv_bn_bwd_from_partials merges)."""
    from ssv_amd import ops
    x, w, dy, aff = _case(dev, n, h, w_, c, k, seed=10)
    refdx = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    wd, dyd = w.contiguous(memory_format=torch.channels_last).to(dev), dy.to(dev)
    dx = ops.wino_conv2d_dgrad(dyd, wd)
    assert rel(dx, refdx) < 2e-6, rel(dx, refdx)
    gx = seeded_randn(20, n, h, w_, c)
    mean, invstd = seeded_randn(21, c) * 0.1, seeded_randn(22, c).abs() + 0.5
    bit = (gx * aff[0] + aff[1]) > 0                                   # fp32, like the kernel (fmaf vs mul+add can differ on a tie: none in this data)
    gate = ops.BnGateCtx(gx.to(dev), mean.to(dev), invstd.to(dev), scale=aff[0].to(dev), shift=aff[1].to(dev))
    g = ops.wino_conv2d_dgrad(dyd, wd, gate=gate)
    want = torch.where(bit, refdx, torch.zeros_like(refdx))
    near = ((gx * aff[0] + aff[1]).abs() < 1e-6)                       # elements whose gate bit is decided by rounding
    assert int(near.sum()) < 4
    diff = (g.cpu().double() - want).abs()
    diff[near] = 0
    assert float(diff.max()) < 2e-5 * float(refdx.abs().max())
    # the same gate from the forward's byte mask (what a materialised activation carries): identical bits
    mask = torch.zeros(n * h * w_ * c // 4, dtype=torch.uint8)
    bits = bit.reshape(-1, 4).to(torch.uint8)
    mask = (bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3)).to(torch.uint8)
    g2 = ops.wino_conv2d_dgrad(dyd, wd, gate=ops.BnGateCtx(gx.to(dev), mean.to(dev), invstd.to(dev), mask=mask.to(dev)))
    if int(near.sum()) == 0:
        assert torch.equal(g2, g) and torch.equal(g2._gate_partials[0], g._gate_partials[0]) and torch.equal(g2._gate_partials[1], g._gate_partials[1])
    sg, sgx, groups = g._gate_partials
    assert groups == sg.shape[0] == (n * ((h + 1) // 2) * ((w_ + 1) // 2) + 15) // 16
    xhat = (gx.double() - mean.double()) * invstd.double()
    np.testing.assert_allclose(sg.sum(0).cpu().double().numpy(), want.reshape(-1, c).sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(want.abs().sum(dim=(0, 1, 2)).max()))
    np.testing.assert_allclose(sgx.sum(0).cpu().double().numpy(), (want * xhat).reshape(-1, c).sum(0).numpy(), rtol=1e-4, atol=1e-4 * float((want * xhat).abs().sum(dim=(0, 1, 2)).max()))


@pytest.mark.parametrize("n,h,w_,c,k", SHAPES)
def test_winograd_weight_gradient_from_the_kept_transformed_input(dev, n, h, w_, c, k):
    from ssv_amd import ops
    x, w, dy, aff = _case(dev, n, h, w_, c, k, seed=30)
    a = torch.relu(x.double() * aff[0].double() + aff[1].double())
    wr = w.double().clone().requires_grad_()
    F.conv2d(a.permute(0, 3, 1, 2), wr, padding=1).backward(dy.double().permute(0, 3, 1, 2))
    xd, wd, dyd = x.to(dev), w.contiguous(memory_format=torch.channels_last).to(dev), dy.to(dev)
    _, _, v = ops.wino_conv2d_fwd(xd, wd, in_affine=(aff[0].to(dev), aff[1].to(dev)), keep_v=True)
    dw = torch.full_like(wd, 0.25)
    ops.wino_conv2d_wgrad(v, dyd, wd, dw, accumulate=True)            # accumulates into what is there
    assert rel(dw - 0.25, wr.grad) < 5e-6, rel(dw - 0.25, wr.grad)
    dw2 = torch.full_like(wd, 7.0)
    ops.wino_conv2d_wgrad(v, dyd, wd, dw2, accumulate=False)
    assert rel(dw2, wr.grad) < 2e-6


def test_deep_3x3_layers_take_winograd_and_a_bottleneck_matches_the_direct_kernels(dev):
    """The dispatch rule (ops.use_winograd) on the ResNet-50 @224 shapes, and one stage-3 bottleneck unit forward + backward through
    Winograd against the same unit on the direct kernels (SSV_NO_WINOGRAD's switch): two fp32 evaluations, rounding-level agreement."""
    from ssv_amd import nn as hnn, ops
    from ssv_amd.networks import resnet
    b = 512
    for hw, ch, want in ((56, 64, False), (28, 128, True), (14, 256, True), (7, 512, True)):
        assert ops.use_winograd((ch, ch, 3, 3), 1, 1, (b, hw, hw, ch), True) is want
    assert not ops.use_winograd((256, 256, 3, 3), 2, 1, (b, 28, 28, 256), True)              # stride 2: direct
    assert not ops.use_winograd((256, 256, 1, 1), 1, 0, (b, 14, 14, 256), True)
    assert not ops.use_winograd((256, 256, 3, 3), 1, 1, (2, 14, 14, 256), True)              # too few tiles
    outs, calls = [], []
    inner = ops.wino_conv2d_fwd
    for wino in (True, False):
        prev, ops.WINOGRAD = ops.WINOGRAD, wino
        calls.append(0)

        def counted(*a, **k):
            calls[-1] += 1
            return inner(*a, **k)
        ops.wino_conv2d_fwd = counted
        try:
            torch.manual_seed(5)
            blk = resnet.Bottleneck(1024, 256).to(dev)
            x = seeded_randn(41, 8, 14, 14, 1024).to(dev).requires_grad_()
            y = blk(x)
            y.backward(seeded_randn(42, 8, 14, 14, 1024).to(dev))
            torch.cuda.synchronize()
            outs.append((y.detach().cpu(), x.grad.cpu(), {k_: p.grad.detach().cpu().clone() for k_, p in blk.named_parameters()}))
        finally:
            ops.WINOGRAD = prev
            ops.wino_conv2d_fwd = inner
    assert calls == [1, 0], calls
    (yw, dxw, gw), (yd, dxd, gd) = outs
    assert rel(yw, yd.double()) < 2e-6
    # backward: the two evaluations' BatchNorm statistics differ in the last bits, so a ReLU within rounding of zero may gate differently
    # (one flip among the unit's 4e5 activations moves a gradient by ~1e-3 relative): size class for the norm, tight for the bulk
    assert rel(dxw, dxd.double()) < 5e-3
    d = (dxw.double() - dxd.double()).abs()
    assert float((d > 1e-4 * float(dxd.abs().max())).double().mean()) < 2e-2
    for k_ in gw:
        assert rel(gw[k_], gd[k_].double()) < 5e-3, (k_, rel(gw[k_], gd[k_].double()))
