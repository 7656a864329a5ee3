"""GPU: 3x3 / stride 1 / padding 1 layers with fewer than 128 output channels (networks/resnet.py:7-10,56-58: conv2 of the 64-channel units, and its data gradient,
which is the same product with the rotated filter) on the 256 x 64 tile of the forward kernel (csrc/conv_mfma.hip): the shipped generic loader and, with SSV_HIP_LIB
pointing at a -DSSV_EXP_HALO diagnostic build (tools/exp/r04_halo.sh), the halo loader (conv_fwd_k with C4 == 3), which stages every pixel once per 16-channel chunk
instead of once per tap.  Against torch fp64 convolutions: plain forward, the statistics epilogue, the data gradient with addend and with the BatchNorm + ReLU gate
(recomputed and byte-mask) and its partial sums; tiles that start mid-row, span two images, end past M; maps too wide / too narrow for the halo loader's staged
records.  Tolerances are those of tests/test_gpu_ops.py."""
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


def close(got, ref, rtol=1e-4, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = float(ref.abs().max()) + 1e-30
    err = (got - ref).abs()
    bad = err > 2e-5 * scale + rtol * ref.abs()
    assert not bool(bad.any()), f"{what}: max err {float(err.max()):.3e} (scale {scale:.3e}), {int(bad.sum())}/{bad.numel()} out of tolerance"


# n, h, w, c, k: 56x56 (a tile = 4.57 rows: every start column, tiles across two images), 32x32 (tiles are whole rows), three chunks, ragged last tile,
# non-square maps, K below one column tile, a 3-wide map (86 staged rows), and maps the records do not hold (w = 2, w = 96: generic loader)
CASES = [(3, 56, 56, 64, 64), (2, 32, 32, 64, 64), (5, 8, 8, 32, 64), (2, 16, 16, 96, 64), (1, 64, 64, 64, 64), (3, 7, 9, 64, 48), (7, 5, 3, 32, 64), (9, 6, 2, 32, 64),
         (1, 4, 96, 32, 64), (1, 1, 1, 64, 64)]


@pytest.mark.parametrize("n,h,w_,c,k", CASES)
def test_forward_plain_and_with_statistics(dev, n, h, w_, c, k):
    from ssv_amd import ops
    x = seeded_randn(1, n, h, w_, c)
    w = seeded_randn(2, k, c, 3, 3) * (2.0 / (9 * c)) ** 0.5
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    xd, wd = x.to(dev), w.contiguous(memory_format=torch.channels_last).to(dev)
    y = ops.conv2d_fwd(xd, wd, 1, 1)
    close(y, ref, what="forward")
    rel = float((y.cpu().double() - ref).norm() / ref.norm())
    assert rel < 1e-6, rel
    bias, add = seeded_randn(3, k).to(dev), seeded_randn(4, n, h, w_, k).to(dev)
    close(ops.conv2d_fwd(xd, wd, 1, 1, bias=bias, addend=add), ref + bias.cpu().double() + add.cpu().double(), what="forward + bias + addend")
    got = ops.conv2d_fwd_stats(xd, wd, 1, 1)
    if got is None:
        return
    ys, pmean, pm2 = got
    assert torch.equal(ys, y)
    m = n * h * w_
    y2 = y.view(m, k).cpu().double()
    assert pmean.shape == ((m + 63) // 64, k)
    for g in range(pmean.shape[0]):
        blk = y2[64 * g:64 * g + 64]
        np.testing.assert_allclose(pmean[g].cpu().numpy(), blk.mean(0).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(pm2[g].cpu().numpy(), ((blk - blk.mean(0)) ** 2).sum(0).numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("n,h,w_,c,k", CASES)
def test_data_gradient_with_addend_and_gate(dev, n, h, w_, c, k):
    """The data gradient of the same layers runs on the forward kernel with the rotated filter: contraction over k, c output columns."""
    from ssv_amd import ops
    if c % 4 or k % 32:
        pytest.skip("the data gradient contracts over K: needs K % 32 == 0 for this path")
    w = seeded_randn(2, k, c, 3, 3) * (2.0 / (9 * k)) ** 0.5
    dy = seeded_randn(3, n, h, w_, k)
    refdx = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    wd, dyd = w.contiguous(memory_format=torch.channels_last).to(dev), dy.to(dev)
    shape = (n, h, w_, c)
    dx = ops.conv2d_dgrad(dyd, wd, shape, 1, 1)
    close(dx, refdx, what="data gradient")
    add = seeded_randn(4, *shape)
    buf = add.to(dev).clone()
    close(ops.conv2d_dgrad(dyd, wd, shape, 1, 1, addend=buf, out=buf), refdx + add.double(), what="data gradient + addend")
    gx = seeded_randn(20, *shape)
    scale, shift = seeded_randn(4, c) * 0.3 + 1.0, seeded_randn(5, c) * 0.2
    mean, invstd = seeded_randn(21, c) * 0.1, seeded_randn(22, c).abs() + 0.5
    bit = (gx * scale + shift) > 0
    near = (gx * scale + shift).abs() < 1e-6
    g = ops.conv2d_dgrad(dyd, wd, shape, 1, 1, gate=ops.BnGateCtx(gx.to(dev), mean.to(dev), invstd.to(dev), scale=scale.to(dev), shift=shift.to(dev)))
    want = torch.where(bit, refdx, torch.zeros_like(refdx))
    diff = (g.cpu().double() - want).abs()
    diff[near] = 0
    assert float(diff.max()) < 4e-5 * float(refdx.abs().max())
    bits = bit.reshape(-1, 4).to(torch.uint8)
    mask = (bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3)).to(torch.uint8)
    g2 = ops.conv2d_dgrad(dyd, wd, shape, 1, 1, gate=ops.BnGateCtx(gx.to(dev), mean.to(dev), invstd.to(dev), mask=mask.to(dev)))
    if int(near.sum()) == 0:
        assert torch.equal(g2, g)
    sg, sgx = g._gate_partials[0], g._gate_partials[1]
    xhat = (gx.double() - mean.double()) * invstd.double()
    np.testing.assert_allclose(sg.sum(0).cpu().double().numpy(), want.reshape(-1, c).sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(want.abs().sum(dim=(0, 1, 2)).max()))
    np.testing.assert_allclose(sgx.sum(0).cpu().double().numpy(), (want * xhat).reshape(-1, c).sum(0).numpy(), rtol=1e-4,
                               atol=1e-4 * float((want * xhat).abs().sum(dim=(0, 1, 2)).max()))


def test_repeat_launches_are_bitwise_equal(dev):
    from ssv_amd import ops
    x = seeded_randn(1, 4, 56, 56, 64).to(dev)
    w = (seeded_randn(2, 64, 64, 3, 3) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
    a = ops.conv2d_fwd(x, w, 1, 1)
    for _ in range(3):
        assert torch.equal(ops.conv2d_fwd(x, w, 1, 1), a)
