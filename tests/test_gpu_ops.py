"""GPU parity of every HIP kernel (through the C ABI) against plain torch fp32 CPU ops / the oracle.
Tolerances: fp32 MFMA is an exact fp32 fmaf chain, so differences are summation-order only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import seeded_randn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from ssv_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def nhwc(x):      # CPU NCHW -> CPU NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def close(got, ref, rtol=1e-4, atol=None, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = float(ref.abs().max()) + 1e-30
    atol = 2e-5 * scale if atol is None else atol
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{what}: max err {float(err.max()):.3e} (scale {scale:.3e}), {int(bad.sum())}/{bad.numel()} out of tolerance"


CONV_CASES = [
    # N, H, W, C, K, R, stride, pad
    (2, 8, 8, 64, 64, 1, 1, 0),        # layer1 1x1 (256x64 tile)
    (3, 9, 9, 64, 256, 1, 1, 0),       # ragged M (243 rows)
    (2, 8, 8, 256, 64, 1, 1, 0),
    (2, 8, 8, 64, 64, 3, 1, 1),        # 3x3 s1
    (2, 10, 10, 128, 128, 3, 2, 1),    # 3x3 s2, even input
    (2, 9, 9, 128, 128, 3, 2, 1),      # 3x3 s2, odd input
    (2, 8, 8, 256, 512, 1, 2, 0),      # downsample 1x1 s2
    (1, 14, 14, 32, 48, 3, 1, 1),      # K not a multiple of the tile
    (5, 1, 1, 2048, 128, 1, 1, 0),     # Linear 2048 -> 128, tiny batch
    (64, 1, 1, 512, 512, 1, 1, 0),     # Linear
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case):
    from ssv_amd import ops
    n, h, w, c, k, r, st, pad = case
    x = seeded_randn(1, n, c, h, w)
    wt = seeded_randn(2, k, c, r, r) / (c * r * r) ** 0.5
    x.requires_grad_(), wt.requires_grad_()
    y = F.conv2d(x, wt, stride=st, padding=pad)
    dy = seeded_randn(3, *y.shape)
    y.backward(dy)
    xd, wd, dyd = nhwc(x.detach()).to(dev), wt.detach().contiguous(memory_format=torch.channels_last).to(dev), nhwc(dy).to(dev)
    yd = ops.conv2d_fwd(xd, wd, st, pad)
    close(nchw(yd.cpu()), y, what="conv fwd")
    dxd = ops.conv2d_dgrad(dyd, wd, xd.shape, st, pad)
    close(nchw(dxd.cpu()), x.grad, what="conv dgrad")
    dwd = torch.zeros_like(wd)
    ops.conv2d_wgrad(xd, dyd, wd, dwd, st, pad, accumulate=False)
    close(dwd.cpu(), wt.grad, what="conv wgrad")
    # accumulate + fused addend / bias
    ops.conv2d_wgrad(xd, dyd, wd, dwd, st, pad, accumulate=True)
    close(dwd.cpu(), 2 * wt.grad, what="conv wgrad accumulate")
    add = seeded_randn(4, *xd.shape).to(dev)
    dx2 = ops.conv2d_dgrad(dyd, wd, xd.shape, st, pad, addend=add, out=add)
    close(nchw(dx2.cpu()), x.grad + nchw(seeded_randn(4, *xd.shape)), what="conv dgrad addend")
    bias = seeded_randn(5, k).to(dev)
    yb = ops.conv2d_fwd(xd, wd, st, pad, bias=bias)
    close(nchw(yb.cpu()), y + bias.cpu().view(1, -1, 1, 1), what="conv fwd bias")


@pytest.mark.parametrize("m,c,k", [(300, 384, 1152), (4100, 64, 128), (130, 1536, 384), (97, 128, 48), (37888, 384, 384)])
def test_linear_weight_and_bias_gradient_in_one_pass(dev, m, c, k):
    """nn.Linear backward: dW = dY^T X and db = column sums of dY from ONE pass over dY (ssv_conv2d_wgrad_bias) - against fp64, against the
    stand-alone column-sum kernel, accumulating, and repeatable bit for bit (fixed-order reduction)."""
    from ssv_amd import ops
    x, dy = seeded_randn(51, m, c), seeded_randn(52, m, k)
    w = torch.empty(k, c, 1, 1).contiguous(memory_format=torch.channels_last).to(dev)
    xd, dyd = x.to(dev).view(m, 1, 1, c), dy.to(dev).view(m, 1, 1, k)
    dw, db = torch.zeros_like(w), torch.zeros(k, device=dev)
    assert ops.FUSE_BIAS_GRAD
    ops.conv2d_wgrad(xd, dyd, w, dw, 1, 0, accumulate=False, dbias=db)
    close(dw.view(k, c).cpu(), dy.double().t() @ x.double(), what="dW")
    close(db.cpu(), dy.double().sum(0), what="db")
    ref = torch.zeros(k, device=dev)
    ops.colsum(dyd, ref, accumulate=False)
    close(db, ref.cpu(), rtol=1e-5, what="db vs column-sum kernel")
    dw2, db2 = torch.zeros_like(w), torch.zeros(k, device=dev)
    ops.conv2d_wgrad(xd, dyd, w, dw2, 1, 0, accumulate=False, dbias=db2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    ops.conv2d_wgrad(xd, dyd, w, dw2, 1, 0, accumulate=True, dbias=db2)
    close(db2.cpu(), 2 * dy.double().sum(0), what="db accumulate")
    close(dw2.view(k, c).cpu(), 2 * (dy.double().t() @ x.double()), what="dW accumulate")


GROUPED_CASES = [
    # N, H, W, C, K, groups, stride       (3x3, padding 1: conv3x3(groups = 32) of the ResNeXt bottlenecks, networks/resnet.py:8-10,57)
    (4, 12, 12, 128, 128, 32, 1),      # 32x4d layer1: 4 channels per group
    (3, 10, 10, 256, 256, 32, 2),      # first unit of layer2: stride 2 (parity-class data gradient)
    (2, 7, 7, 512, 512, 32, 1),        # 16 per group
    (2, 6, 6, 1024, 1024, 32, 1),      # 32 per group
    (2, 9, 9, 128, 256, 32, 1),        # 4 in, 8 out per group: the spans differ
    (2, 8, 8, 192, 192, 3, 1),         # groups that straddle the 64-column tiles
]


@pytest.mark.parametrize("case", GROUPED_CASES)
def test_grouped_conv_on_its_block_diagonal_bank(dev, case):
    """The group-aware entry points (every tile contracts over the channels of its own groups only) against torch's grouped convolution in fp64,
    and against the plain dense kernels on the same block-diagonal bank: forward and data gradient bit for bit (the skipped products are
    exact zeros), weight gradient on the diagonal blocks (the only entries the grouped form defines)."""
    from ssv_amd import ops
    n, h, w, c, k, g, st = case
    cg = c // g
    x = seeded_randn(41, n, c, h, w).double().requires_grad_()
    wg = (seeded_randn(42, k, cg, 3, 3) / (cg * 9) ** 0.5).double().requires_grad_()
    y = F.conv2d(x, wg, stride=st, padding=1, groups=g)
    dy = seeded_randn(43, *y.shape).double()
    y.backward(dy)
    xd, dyd = nhwc(x.detach().float()).to(dev), nhwc(dy.float()).to(dev)
    wgd = wg.detach().float().contiguous(memory_format=torch.channels_last).to(dev)
    bank = ops.group_expand(wgd, g)
    assert tuple(bank.shape) == (k, c, 3, 3)
    yg, yd = ops.conv2d_fwd(xd, bank, st, 1, groups=g), ops.conv2d_fwd(xd, bank, st, 1)
    close(nchw(yg.cpu()), y, what="grouped fwd")
    assert torch.equal(yg, yd)
    add = seeded_randn(44, *xd.shape).to(dev)
    dxg = ops.conv2d_dgrad(dyd, bank, xd.shape, st, 1, addend=add, groups=g)
    dxd = ops.conv2d_dgrad(dyd, bank, xd.shape, st, 1, addend=add)
    close(nchw(dxg.cpu()), x.grad + nchw(add.cpu()).double(), what="grouped dgrad")
    if st == 1:
        assert torch.equal(dxg, dxd)
    else:
        # the strided data gradient of a BANK runs its 64-column tile on the fp32 MFMA instruction in either arithmetic (no bf16-piece variant of that tile), the
        # dense product of the same tensor takes the 128-column bf16x3 kernel: bit for bit only where both run the same instruction
        with ops.arithmetic("f32"):
            assert torch.equal(ops.conv2d_dgrad(dyd, bank, xd.shape, st, 1, addend=add, groups=g), ops.conv2d_dgrad(dyd, bank, xd.shape, st, 1, addend=add))
        close(dxg.cpu(), dxd.cpu(), rtol=1e-5, what="grouped vs dense strided dgrad")
    dbank = torch.full_like(bank, float("nan"))             # whatever the skipped tiles leave behind must never be read
    ops.conv2d_wgrad(xd, dyd, bank, dbank, st, 1, accumulate=False, groups=g)
    dwg = torch.zeros_like(wgd)
    ops.group_extract(dbank, dwg, g, accumulate=False)
    close(dwg.cpu(), wg.grad, what="grouped wgrad")
    ddense = torch.zeros_like(bank)
    ops.conv2d_wgrad(xd, dyd, bank, ddense, st, 1, accumulate=False)
    dwg2 = torch.zeros_like(wgd)
    ops.group_extract(ddense, dwg2, g, accumulate=False)
    close(dwg, dwg2.cpu(), rtol=1e-5, what="grouped vs dense wgrad")
    with pytest.raises(Exception):
        ops.conv2d_fwd(xd, bank, st, 1, groups=7)           # groups must divide C and K


@pytest.mark.parametrize("case", [(2, 20, 20, 3, 64, 7, 2, 3), (3, 12, 12, 3, 64, 3, 1, 1), (2, 11, 13, 3, 64, 7, 2, 3)])
def test_stem_conv_generic_gather(dev, case):
    from ssv_amd import ops
    n, h, w, c, k, r, st, pad = case
    x = seeded_randn(6, n, c, h, w)
    wt = (seeded_randn(7, k, c, r, r) / (c * r * r) ** 0.5).requires_grad_()
    y = F.conv2d(x, wt, stride=st, padding=pad)
    dy = seeded_randn(8, *y.shape)
    y.backward(dy)
    xd, wd, dyd = nhwc(x).to(dev), wt.detach().contiguous(memory_format=torch.channels_last).to(dev), nhwc(dy).to(dev)
    close(nchw(ops.conv2d_fwd(xd, wd, st, pad).cpu()), y, what="stem fwd")
    dwd = torch.zeros_like(wd)
    ops.conv2d_wgrad(xd, dyd, wd, dwd, st, pad, accumulate=False)
    close(dwd.cpu(), wt.grad, what="stem wgrad")


@pytest.mark.parametrize("shape,relu,res", [((4, 6, 6, 64), True, False), ((3, 5, 5, 256), True, True), ((2, 7, 7, 128), False, False),
                                             ((37, 1, 1, 2048), True, False), ((16, 1, 1, 128), False, False), ((2, 30, 30, 64), True, True)])
def test_batchnorm_fwd_bwd(dev, shape, relu, res):
    from ssv_amd import ops
    n, h, w, c = shape
    x = (seeded_randn(11, n, c, h, w) * 1.7 + 0.6).requires_grad_()
    gamma = (seeded_randn(12, c) * 0.3 + 1.0).requires_grad_()
    beta = (seeded_randn(13, c) * 0.2).requires_grad_()
    r = seeded_randn(14, n, c, h, w).requires_grad_() if res else None
    rm, rv = torch.zeros(c), torch.ones(c)
    y = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    dy = seeded_randn(15, *y.shape)
    y.backward(dy)
    g = lambda t: t.to(dev)
    rmd, rvd, nbt = g(torch.zeros(c)), g(torch.ones(c)), g(torch.zeros((), dtype=torch.long))
    xd = g(nhwc(x.detach()))
    yd, mean, invstd, mask = ops.bn_train_fwd(xd, g(gamma.detach()), g(beta.detach()), rmd, rvd, nbt, relu=relu,
                                              residual=g(nhwc(r.detach())) if res else None, want_mask=True)
    assert (mask is not None) == relu
    close(nchw(yd.cpu()), y, what="bn fwd")
    close(rmd, rm, what="running_mean")
    close(rvd, rv, what="running_var")
    assert int(nbt.item()) == 1
    dgamma, dbeta = g(torch.zeros(c)), g(torch.zeros(c))
    dxd, dres = ops.bn_train_bwd(g(nhwc(dy)), yd, xd, g(gamma.detach()), mean, invstd, relu, dgamma, dbeta, want_dres=res, accumulate=True)
    close(nchw(dxd.cpu()), x.grad, rtol=2e-4, what="bn dx")
    close(dgamma, gamma.grad, rtol=2e-4, what="bn dgamma")
    close(dbeta, beta.grad, rtol=2e-4, what="bn dbeta")
    if res:
        close(nchw(dres.cpu()), r.grad, what="bn dres")
    if relu:      # same backward from the byte mask instead of y: identical bits
        dg2, db2 = g(torch.zeros(c)), g(torch.zeros(c))
        dx2, dres2 = ops.bn_train_bwd(g(nhwc(dy)), None, xd, g(gamma.detach()), mean, invstd, relu, dg2, db2, want_dres=res, accumulate=True, relu_mask=mask)
        assert torch.equal(dx2, dxd) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)
        if res:
            assert torch.equal(dres2, dres)


def test_pools_and_layout(dev):
    from ssv_amd import ops
    for (n, c, h, w) in ((2, 64, 12, 12), (3, 64, 9, 11), (1, 8, 2, 2)):
        x = seeded_randn(21, n, c, h, w).relu().requires_grad_()        # ReLU'd input -> ties at zero, like the real net
        y = F.max_pool2d(x, 3, 2, 1)
        dy = seeded_randn(22, *y.shape)
        y.backward(dy)
        xd = nhwc(x.detach()).to(dev)
        yd, am = ops.maxpool_fwd(xd)
        close(nchw(yd.cpu()), y, rtol=0, atol=0, what="maxpool fwd")
        dxd = ops.maxpool_bwd(nhwc(dy).to(dev), am, xd.shape)
        mask = (x.detach() > 0).float()                                   # zero-valued ties are killed by the ReLU mask upstream
        close(nchw(dxd.cpu()) * mask, x.grad * mask, what="maxpool bwd")
        close(dxd.sum(), x.grad.sum(), rtol=1e-4, what="maxpool bwd mass")
    x = seeded_randn(23, 5, 2048, 7, 7).requires_grad_()
    y = x.mean(dim=(2, 3))
    dy = seeded_randn(24, 5, 2048)
    y.backward(dy)
    xd = nhwc(x.detach()).to(dev)
    close(ops.gap_fwd(xd), y, what="gap fwd")
    close(nchw(ops.gap_bwd(dy.to(dev), xd.shape).cpu()), x.grad, what="gap bwd")
    img = seeded_randn(25, 3, 3, 10, 14)
    close(ops.nchw_to_nhwc(img.to(dev)), nhwc(img), rtol=0, atol=0, what="nchw->nhwc")
    close(ops.nchw_to_nhwc(img.to(dev).contiguous(memory_format=torch.channels_last)), nhwc(img), rtol=0, atol=0, what="channels_last view")


def test_colsum_add_scale_fill(dev):
    from ssv_amd import ops
    x = seeded_randn(31, 333, 128)
    out = torch.ones(128, device=dev)
    ops.colsum(x.to(dev), out, accumulate=True)
    close(out, 1 + x.sum(0), what="colsum")
    a, b = seeded_randn(32, 1001).to(dev), seeded_randn(33, 1001).to(dev)
    ref = a.cpu() + b.cpu()
    close(ops.add_(a, b), ref, rtol=0, atol=0, what="add")
    f = torch.tensor(0.25, device=dev)
    close(ops.scale_(a, f), ref * 0.25, rtol=0, atol=0, what="scale")
    close(ops.fill_(a, 3.0), torch.full((1001,), 3.0), rtol=0, atol=0, what="fill")


def test_sgd_and_ema_match_reference(dev, golden):
    from ssv_amd import _lib
    g = golden["optim_level"]
    shapes = [g["sgd_p0_init"].shape, g["sgd_p1_init"].shape]
    flat = torch.cat([torch.tensor(g["sgd_p0_init"]).flatten(), torch.tensor(g["sgd_p1_init"]).flatten()]).to(dev)
    buf = torch.zeros_like(flat)
    n0 = int(np.prod(shapes[0]))
    for s in range(3):
        grads = torch.cat([seeded_randn(50 + 10 * s + i, *shp).flatten() for i, shp in enumerate(shapes)]).to(dev)
        half = (grads * 0.25).contiguous()                       # split the gradient over the two slabs: g + g2
        rest = (grads - half).contiguous()
        _lib.call("ssv_sgd_nesterov", flat.numel(), _lib.ptr(flat), _lib.ptr(rest), _lib.ptr(half) if s else 0, _lib.ptr(buf), 0.3, 1e-2, 0.9, int(s == 0), _lib.stream()) if s else \
            _lib.call("ssv_sgd_nesterov", flat.numel(), _lib.ptr(flat), _lib.ptr(grads), 0, _lib.ptr(buf), 0.3, 1e-2, 0.9, 1, _lib.stream())
        close(flat[:n0].view(shapes[0]), torch.tensor(g[f"sgd_p0_step{s}"]), rtol=2e-6, atol=2e-7, what=f"sgd p0 step {s}")
        close(flat[n0:], torch.tensor(g[f"sgd_p1_step{s}"]), rtol=2e-6, atol=2e-7, what=f"sgd p1 step {s}")
    t, o = seeded_randn(41, 777), seeded_randn(42, 777)
    td = t.to(dev)
    _lib.call("ssv_ema", 777, _lib.ptr(td), _lib.ptr(o.to(dev)), 0.996, _lib.stream())
    close(td, 0.996 * t + (1.0 - 0.996) * o, rtol=1e-6, what="ema")


def test_l2norm(dev):
    from ssv_amd import ops
    z = seeded_randn(51, 37, 100).requires_grad_()
    zh = F.normalize(z, p=2, dim=-1)
    d = seeded_randn(52, 37, 100)
    zh.backward(d)
    zd = z.detach().to(dev)
    zhat, inv = ops.l2norm_fwd(zd, True, ldo=128)
    close(zhat[:, :100], zh, what="l2norm fwd")
    assert float(zhat[:, 100:].abs().max()) == 0.0
    dpad = torch.zeros(37, 128)
    dpad[:, :100] = d
    close(ops.l2norm_bwd(zhat, inv, dpad.to(dev), 100, True), z.grad, what="l2norm bwd")


def test_ntxent_matches_reference_golden(dev, golden):
    from ssv_amd.utils import losses
    g = golden["loss_level"]
    for tag, (n, d, seed, norm, temp) in zip("abcde", g["ntxent_cases"]):
        n, d, seed = int(n), int(d), int(seed)
        zi = seeded_randn(seed, n, d).to(dev).requires_grad_()
        zj = seeded_randn(seed + 1000, n, d).to(dev).requires_grad_()
        loss = losses.SimclrLoss(bool(norm), float(temp))(zi, zj)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"ntxent_{tag}_loss"], rtol=5e-6, err_msg=f"case {tag}")
        close(zi.grad, torch.tensor(g[f"ntxent_{tag}_dzi"]), rtol=2e-4, what=f"ntxent dzi {tag}")
        close(zj.grad, torch.tensor(g[f"ntxent_{tag}_dzj"]), rtol=2e-4, what=f"ntxent dzj {tag}")


def test_ntxent_large_and_scaled_upstream(dev):
    """N=256 (BASELINE config 2 loss shape) against the oracle, with a non-unit upstream gradient."""
    from ssv_amd.utils import losses
    zi, zj = seeded_randn(61, 256, 128), seeded_randn(62, 256, 128)
    a, b = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    ref = oracle.ntxent_loss(a, b, True, 0.5)
    (ref * 0.37).backward()
    zid, zjd = zi.to(dev).requires_grad_(), zj.to(dev).requires_grad_()
    loss = losses.SimclrLoss(True, 0.5)(zid, zjd)
    (loss * 0.37).backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
    close(zid.grad, a.grad, rtol=2e-4, what="dzi")
    close(zjd.grad, b.grad, rtol=2e-4, what="dzj")


def test_ntxent_sharded_rows_equal_global(dev):
    """Multi-GPU form on one device: each 'rank' computes its own row block against the gathered Z;
    the per-rank losses sum to the global loss and the per-rank dZ blocks tile the global dZ."""
    from ssv_amd import _lib, ops
    nglob, d, world = 96, 64, 4
    b = nglob // world
    zi, zj = seeded_randn(71, nglob, d), seeded_randn(72, nglob, d)
    a, c = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    ref = oracle.ntxent_loss(a, c, True, 0.2)
    ref.backward()
    zall = torch.cat([oracle.l2_normalize(zi), oracle.l2_normalize(zj)]).to(dev).contiguous()
    lse_all = torch.empty(2 * nglob, device=dev)
    total = 0.0
    parts = []
    for r in range(world):
        lse, pos = torch.empty(2 * b, device=dev), torch.empty(2 * b, device=dev)
        _lib.call("ssv_ntxent_fwd", nglob, b, r * b, d, _lib.ptr(zall), 5.0, _lib.ptr(lse), _lib.ptr(pos), _lib.stream())
        out = torch.empty((), device=dev)
        _lib.call("ssv_ntxent_loss", 2 * b, _lib.ptr(lse), _lib.ptr(pos), 1.0 / (2 * nglob), _lib.ptr(out), _lib.stream())
        total += out.item()
        lse_all[r * b:(r + 1) * b] = lse[:b]
        lse_all[nglob + r * b:nglob + (r + 1) * b] = lse[b:]
    np.testing.assert_allclose(total, ref.item(), rtol=3e-6)
    # reference dZhat from autograd on normalised inputs
    zh_i, zh_j = oracle.l2_normalize(zi).detach().requires_grad_(), oracle.l2_normalize(zj).detach().requires_grad_()
    oracle.ntxent_loss(zh_i, zh_j, False, 0.2).backward()
    for r in range(world):
        dz = torch.empty((2 * b, d), device=dev)
        _lib.call("ssv_ntxent_bwd", nglob, b, r * b, d, _lib.ptr(zall), _lib.ptr(lse_all), 5.0, 5.0 / (2 * nglob), _lib.ptr(dz), _lib.stream())
        close(dz[:b], zh_i.grad[r * b:(r + 1) * b], rtol=2e-4, what=f"rank {r} dzi")
        close(dz[b:], zh_j.grad[r * b:(r + 1) * b], rtol=2e-4, what=f"rank {r} dzj")


def test_byol_loss_matches_reference_golden(dev, golden):
    from ssv_amd import ops
    from ssv_amd.utils import losses
    g = golden["loss_level"]
    p1, p2 = seeded_randn(31, 16, 128), seeded_randn(32, 16, 128)
    t1, t2 = oracle.l2_normalize(seeded_randn(33, 16, 128)).to(dev), oracle.l2_normalize(seeded_randn(34, 16, 128)).to(dev)
    h1, i1 = ops.l2norm_fwd(p1.to(dev), True)
    h2, i2 = ops.l2norm_fwd(p2.to(dev), True)
    o1, o2 = h1.clone().requires_grad_(), h2.clone().requires_grad_()
    loss = losses.byol_pair_loss(o1, o2, t1, t2)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["byol_loss"], rtol=3e-6)
    close(ops.l2norm_bwd(h1, i1, o1.grad, 128, True), torch.tensor(g["byol_dp1"]), rtol=2e-4, what="byol dp1")
    close(ops.l2norm_bwd(h2, i2, o2.grad, 128, True), torch.tensor(g["byol_dp2"]), rtol=2e-4, what="byol dp2")


def test_barlow_matches_reference_golden(dev, golden):
    from ssv_amd.utils import losses
    g = golden["loss_level"]
    for tag, (b, d, seed, norm, lm) in zip("abc", g["barlow_cases"]):
        b, d, seed = int(b), int(d), int(seed)
        zi = seeded_randn(seed, b, d).to(dev).requires_grad_()
        zj = seeded_randn(seed + 1000, b, d).to(dev).requires_grad_()
        loss = losses.BarlowLoss(bool(norm), float(lm))(zi, zj)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"barlow_{tag}_loss"], rtol=2e-5, err_msg=f"case {tag}")
        close(zi.grad, torch.tensor(g[f"barlow_{tag}_dzi"]), rtol=5e-4, what=f"barlow dzi {tag}")
        close(zj.grad, torch.tensor(g[f"barlow_{tag}_dzj"]), rtol=5e-4, what=f"barlow dzj {tag}")


def test_barlow_config_shape_matches_oracle(dev):
    """configs/barlow.yaml shape: D = 4096 (a 4096 x 4096 x B contraction on MFMA), B = 128, normalize=False."""
    from ssv_amd.utils import losses
    zi, zj = oracle.l2_normalize(seeded_randn(81, 128, 4096)), oracle.l2_normalize(seeded_randn(82, 128, 4096))   # head ends in L2-normalise
    a, b = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    ref = oracle.barlow_loss(a, b, False, 0.005)
    ref.backward()
    zid, zjd = zi.to(dev).requires_grad_(), zj.to(dev).requires_grad_()
    loss = losses.BarlowLoss(False, 0.005)(zid, zjd)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-5)
    close(zid.grad, a.grad, rtol=1e-3, what="dzi")
    close(zjd.grad, b.grad, rtol=1e-3, what="dzj")


def test_ntxent_global_batch_4096(dev):
    """BASELINE config 3 loss shape: N = 4096 per view (8192 x 8192 x 128 Gram), one rank computing every row."""
    from ssv_amd.utils import losses
    zi, zj = seeded_randn(91, 4096, 128), seeded_randn(92, 4096, 128)
    a, b = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    ref = oracle.ntxent_loss(a, b, True, 0.5)
    ref.backward()
    zid, zjd = zi.to(dev).requires_grad_(), zj.to(dev).requires_grad_()
    loss = losses.SimclrLoss(True, 0.5)(zid, zjd)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
    close(zid.grad, a.grad, rtol=5e-4, what="dzi")
    close(zjd.grad, b.grad, rtol=5e-4, what="dzj")


def test_filter_transpose_and_stride1_dgrad_route(dev):
    """ssv_filter_transpose, and the stride-1 dgrad computed as a forward convolution with the transposed / rotated filter, against
    the dgrad kernel itself and torch fp64."""
    import torch.nn.functional as F
    from ssv_amd import _lib, ops
    from conftest import seeded_randn
    for (n, h, c, k, r, pad) in ((3, 9, 8, 32, 3, 1), (2, 7, 20, 16, 1, 0), (2, 6, 12, 48, 3, 0), (5, 1, 64, 32, 1, 0)):
        w = (seeded_randn(1, k, c, r, r) * 0.2).contiguous(memory_format=torch.channels_last).to(dev)
        wt = torch.empty((c, r, r, k), device=dev)
        _lib.call("ssv_filter_transpose", k, r, r, c, _lib.ptr(w), _lib.ptr(wt), _lib.stream())
        want = w.permute(0, 2, 3, 1).flip(1, 2).permute(3, 1, 2, 0)                        # [K,R,S,C] -> rotate -> [C,R,S,K]
        assert torch.equal(wt, want.contiguous())
        ho = h + 2 * pad - r + 1
        dy = seeded_randn(2, n, ho, ho, k).to(dev)
        add = seeded_randn(3, n, h, h, c).to(dev)
        got = ops.conv2d_dgrad(dy, w, (n, h, h, c), 1, pad, addend=add)                   # routed through the forward kernel
        d = ops.conv_desc((n, h, h, c), (k, c, r, r), 1, pad)
        ref = torch.empty_like(got)
        _lib.call("ssv_conv2d_dgrad", ops.C.byref(d), _lib.ptr(dy), _lib.ptr(w), _lib.ptr(add), _lib.ptr(ref), _lib.stream())
        x64 = torch.zeros(n, c, h, h, dtype=torch.float64, requires_grad=True)
        F.conv2d(x64, w.cpu().double(), padding=pad).backward(dy.cpu().double().permute(0, 3, 1, 2))
        truth = x64.grad.permute(0, 2, 3, 1) + add.cpu().double()
        np.testing.assert_allclose(got.cpu().double().numpy(), truth.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ref.cpu().double().numpy(), truth.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,h,c,k,r,stride", [(3, 9, 32, 128, 3, 1), (2, 7, 64, 64, 1, 1), (5, 10, 32, 260, 3, 2), (1, 5, 96, 16, 1, 1), (7, 3, 128, 512, 1, 1)])
def test_conv_epilogue_bn_statistics(dev, n, h, c, k, r, stride):
    """ssv_conv2d_fwd_stats: same output as the plain forward, and BatchNorm from its partials == BatchNorm with its own statistics
    pass (ragged row counts: the last 64-row group is partial; K = 260 leaves a ragged column tile)."""
    from ssv_amd import ops
    pad = r // 2
    x = seeded_randn(1, n, h, h, c).to(dev)
    w = (seeded_randn(2, k, c, r, r) * 0.1).contiguous(memory_format=torch.channels_last).to(dev)
    y_ref = ops.conv2d_fwd(x, w, stride, pad)
    y, pmean, pm2 = ops.conv2d_fwd_stats(x, w, stride, pad)
    assert torch.equal(y, y_ref)
    m = y.numel() // k
    assert pmean.shape == ((m + 63) // 64, k)
    y2 = y.view(m, k).cpu().double()
    for g in range(pmean.shape[0]):
        blk = y2[64 * g:64 * g + 64]
        np.testing.assert_allclose(pmean[g].cpu().numpy(), blk.mean(0).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(pm2[g].cpu().numpy(), ((blk - blk.mean(0)) ** 2).sum(0).numpy(), rtol=1e-3, atol=1e-4)
    gamma, beta = (seeded_randn(3, k).abs() + 0.5).to(dev), seeded_randn(4, k).to(dev)
    res = seeded_randn(5, *y.shape).to(dev)
    outs = []
    for partials in (None, (pmean, pm2)):
        rm, rv, nbt = torch.zeros(k, device=dev), torch.ones(k, device=dev), torch.zeros((), dtype=torch.long, device=dev)
        o = ops.bn_train_fwd(y, gamma, beta, rm, rv, nbt, relu=True, residual=res, want_mask=True, partials=partials)
        outs.append((o, rm, rv, nbt))
    (a, arm, arv, anbt), (b, brm, brv, bnbt) = outs
    np.testing.assert_allclose(b[0].cpu().numpy(), a[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b[1].cpu().numpy(), a[1].cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b[2].cpu().numpy(), a[2].cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(brm.cpu().numpy(), arm.cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(brv.cpu().numpy(), arv.cpu().numpy(), rtol=1e-5)
    assert int(anbt) == int(bnbt) == 1
    assert ops.conv2d_fwd_stats(seeded_randn(6, 1, 4, 4, 16).to(dev), (seeded_randn(7, 8, 16, 1, 1)).contiguous(memory_format=torch.channels_last).to(dev)) is None


@pytest.mark.parametrize("n,h,k,r,stride,pad", [(3, 20, 64, 7, 2, 3), (2, 9, 64, 3, 1, 1), (2, 13, 132, 7, 2, 3), (1, 7, 16, 3, 1, 1)])
def test_stem_conv_with_channels_padded_to_four(dev, n, h, k, r, stride, pad):
    """The C == 4 tap-vector gather (forward) and the float4 wgrad on a 3-channel image padded to 4: against torch fp64."""
    from ssv_amd import ops
    x3 = seeded_randn(1, n, h, h, 3)
    w3 = seeded_randn(2, k, 3, r, r) * 0.2
    xr, wr = x3.permute(0, 3, 1, 2).double(), w3.double().requires_grad_()
    ref = F.conv2d(xr, wr, stride=stride, padding=pad)
    dy = seeded_randn(3, *ref.permute(0, 2, 3, 1).shape)
    ref.backward(dy.permute(0, 3, 1, 2).double())
    xp = ops.pad_channels(x3.to(dev), 4)
    assert xp.shape == (n, h, h, 4) and float(xp[..., 3].abs().max()) == 0.0 and torch.equal(xp[..., :3].cpu(), x3)
    wp = ops.pad_channels(w3.permute(0, 2, 3, 1).contiguous().to(dev), 4).permute(0, 3, 1, 2)
    y = ops.conv2d_fwd(xp, wp, stride, pad)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=2e-5)
    dwp = torch.empty_like(wp)
    ops.conv2d_wgrad(xp, dy.to(dev), wp, dwp, stride, pad, accumulate=False)
    dw3 = torch.full((k, r, r, 3), 1.0, device=dev)
    ops.unpad_channels(dwp.permute(0, 2, 3, 1), dw3, accumulate=True)
    np.testing.assert_allclose(dw3.cpu().double().numpy() - 1.0, wr.grad.permute(0, 2, 3, 1).numpy(), rtol=2e-4, atol=2e-4)
    assert float(dwp.permute(0, 2, 3, 1)[..., 3].abs().max()) == 0.0                    # the padded input channel is all zeros


@pytest.mark.parametrize("n,d,norm,temp", [(24, 256, True, 0.5), (21, 200, True, 0.2), (40, 512, False, 1.0)])
def test_ntxent_wide_embeddings_run_through_the_gemm_kernels(dev, n, d, norm, temp):
    """proj_dim > 128 (the reference's SimclrLoss takes any D): the Gram block comes from the MFMA GEMM kernel, lse / weights from
    the row kernels, dZ from a second GEMM - against the oracle, with a non-unit upstream gradient and a row count (2N = 42) that is
    padded to the GEMM's multiple of 16."""
    from ssv_amd.utils import losses
    zi, zj = seeded_randn(91, n, d) * (1.0 if norm else 0.2), seeded_randn(92, n, d) * (1.0 if norm else 0.2)
    a, b = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    ref = oracle.ntxent_loss(a, b, norm, temp)
    (ref * 1.7).backward()
    zid, zjd = zi.to(dev).requires_grad_(), zj.to(dev).requires_grad_()
    loss = losses.SimclrLoss(norm, temp)(zid, zjd)
    (loss * 1.7).backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=5e-6)
    close(zid.grad, a.grad, rtol=2e-4, what="dzi")
    close(zjd.grad, b.grad, rtol=2e-4, what="dzj")


def test_batches_beyond_the_per_launch_limit_are_split_on_the_host(dev, monkeypatch):
    """The conv kernels address a tensor through 32-bit byte offsets (< 2 GiB per launch); ops.* split larger batches along N.  With
    the limit lowered to a few thousand elements every path splits: forward (+ statistics partials, + fused input), data gradient
    (+ gate), weight gradient.  Forward, partials and the gated gradient are bit-identical to the unsplit launch (same per-row
    arithmetic, 64-row partial groups aligned to the chunks); the weight gradient sums its chunks in a different order."""
    from ssv_amd import ops
    n, h, c, k = 24, 8, 32, 64
    x = seeded_randn(11, n, h, h, c).to(dev)
    w = (seeded_randn(12, k, c, 3, 3) * 0.1).contiguous(memory_format=torch.channels_last).to(dev)
    aff = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1)
    dy = seeded_randn(13, n, h, h, k).to(dev)
    gx = seeded_randn(14, n, h, h, c).to(dev)
    mean, invstd = torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5
    mask = torch.randint(0, 16, (x.numel() // 4,), device=dev, dtype=torch.uint8)

    def run():
        y, part = ops.conv2d_fwd_fused(x, w, 1, 1, in_affine=aff, want_stats=True)
        y2 = ops.conv2d_fwd(x, w, 1, 1)
        dx = ops.conv2d_dgrad(dy, w, x.shape, 1, 1, addend=gx.clone(), gate=ops.BnGateCtx(gx, mean, invstd, mask=mask))
        sums = (dx._gate_partials[0].sum(0), dx._gate_partials[1].sum(0))
        dx2 = ops.conv2d_dgrad(dy, w, x.shape, 1, 1, gate=ops.BnGateCtx(gx, mean, invstd, scale=aff[0], shift=aff[1]))
        dw = torch.zeros_like(w)
        ops.conv2d_wgrad(x, dy, w, dw, 1, 1, accumulate=True, in_affine=aff)
        return y, part[0], part[1], y2, dx.clone(), dx2.clone(), sums, dw
    # the operands formed on load (1x1 / stride 1): closing activation formed and written by the convolution, BatchNorm-backward dY with a
    # gate that also reduces against a second BatchNorm input
    w1 = (seeded_randn(15, k, c, 1, 1) * 0.1).contiguous(memory_format=torch.channels_last).to(dev)
    res = seeded_randn(16, n, h, h, c).to(dev)
    xb = seeded_randn(17, n, h, h, k).to(dev)
    coef = (torch.rand(4, k, device=dev) + 0.25).contiguous()
    gx2 = seeded_randn(18, n, h, h, c).to(dev)

    def run_formed():
        y, part, a, am = ops.conv2d_fwd_sumin(x, res, aff[0], aff[1], (aff[0] * 0.5, aff[1] + 0.1), w1, want_mask=True)
        lazy = ops.LazyGrad(dy, xb, coef)
        dx = ops.conv2d_dgrad(lazy, w1, x.shape, 1, 0, addend=gx.clone(), gate=ops.BnGateCtx(gx, mean, invstd, mask=mask, second=(gx2, mean * 0.5, invstd)))
        sums = (dx._gate_partials[0].sum(0), dx._gate_partials[1].sum(0), dx._gate_partials_res[1].sum(0))
        dw = torch.zeros_like(w1)
        ops.conv2d_wgrad(x, lazy, w1, dw, 1, 0, accumulate=True)
        return y, part[0], part[1], a, am, dx.clone(), sums, dw
    whole = run()
    whole_f = run_formed()
    monkeypatch.setattr(ops, "_MAX_ELEMS", 8 * h * h * k + 1)          # 8 samples per launch -> 3 chunks of 8 (8*64 rows: multiple of 64)
    assert len(ops._batch_chunks(n, (h * h * c, h * h * k), rows_per_sample=h * h)) == 3
    split = run()
    split_f = run_formed()
    for i in (0, 1, 2, 3, 4, 5):
        assert torch.equal(whole[i], split[i]), i
    for a, b in zip(whole[6], split[6]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-4)
    close(split[7], whole[7], rtol=1e-5, what="wgrad over chunks")
    for i in (0, 1, 2, 3, 4, 5):
        assert torch.equal(whole_f[i], split_f[i]), f"formed operands, output {i}"
    for a, b in zip(whole_f[6], split_f[6]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-4)
    close(split_f[7], whole_f[7], rtol=1e-5, what="wgrad from the lazy dY over chunks")


@pytest.mark.parametrize("n,h,cin,k,xf,gate_kind", [(3, 7, 64, 256, False, None), (2, 9, 64, 256, True, "affine"), (3, 7, 256, 512, False, "mask"),
                                                   (5, 5, 128, 512, True, None), (2, 6, 512, 2048, False, "affine"),
                                                   (3, 7, 256, 64, False, "mask"), (2, 9, 64, 64, False, None)])
def test_batchnorm_backward_formed_by_the_consumers_of_dx(dev, n, h, cin, k, xf, gate_kind):
    """ops.LazyGrad: conv3 / projection-shortcut gradients (networks/resnet.py:66-75 backwards) take the BatchNorm backward's dx =
    gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)) as (g, x, coefficients) and form it while they stage it; the result must be
    the weight / data gradient computed from the materialised dx (ssv_bn_bwd_from_partials), ragged row counts, fused inputs and gated
    outputs included."""
    from ssv_amd import ops
    m = n * h * h
    g = seeded_randn(901, n, h, h, k).to(dev)
    x = (seeded_randn(902, n, h, h, k) * 1.7 + 0.6).to(dev)                     # the BatchNorm's input (conv output), mean far from 0
    gamma = (torch.rand(k, generator=torch.Generator().manual_seed(5)) + 0.5).to(dev)
    xf64 = x.double().reshape(m, k)
    mean64, var64 = xf64.mean(0), xf64.var(0, unbiased=False)
    mean, invstd = mean64.float(), (1.0 / torch.sqrt(var64 + 1e-5)).float()
    xhat = (xf64 - mean64) / torch.sqrt(var64 + 1e-5)
    g64 = g.double().reshape(m, k)
    groups = -(-m // 64)
    pg = torch.zeros(groups, k, dtype=torch.float64, device=dev)
    pgx = torch.zeros_like(pg)
    for i in range(groups):
        pg[i] = g64[64 * i:64 * i + 64].sum(0)
        pgx[i] = (g64[64 * i:64 * i + 64] * xhat[64 * i:64 * i + 64]).sum(0)
    part = (pg.float().contiguous(), pgx.float().contiguous(), groups)
    dga, dba, dgb, dbb = (torch.zeros(k, device=dev) for _ in range(4))
    dx = ops.bn_bwd_from_partials(g, x, gamma, mean, invstd, part, dga, dba, accumulate=False)
    coef = ops.bn_bwd_coef(x, gamma, mean, invstd, part, dgb, dbb, accumulate=False)
    assert torch.equal(dga, dgb) and torch.equal(dba, dbb)
    want = gamma.double() * invstd.double() * (g64 - g64.mean(0) - xhat * (g64 * xhat).mean(0))
    close(dx.reshape(m, k), want.float(), rtol=1e-4, what="materialised dx")
    lazy = ops.LazyGrad(g, x, coef)
    # the convolution behind the BatchNorm: 1x1, cin -> k
    w = (seeded_randn(903, k, cin, 1, 1) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    xin = seeded_randn(904, n, h, h, cin).to(dev)
    aff = None
    if xf:
        aff = ((torch.rand(cin, generator=torch.Generator().manual_seed(6)) + 0.5).to(dev), seeded_randn(905, cin).to(dev) * 0.3)
    dw1, dw2 = torch.zeros_like(w), torch.zeros_like(w)
    ops.conv2d_wgrad(xin, dx, w, dw1, 1, 0, accumulate=False, in_affine=aff)
    ops.conv2d_wgrad(xin, lazy, w, dw2, 1, 0, accumulate=False, in_affine=aff)
    close(dw2, dw1, rtol=2e-5, what="weight gradient from the lazy dx")
    addend = seeded_randn(906, n, h, h, cin).to(dev)
    gate = None
    if gate_kind is not None:
        gx = seeded_randn(907, n, h, h, cin).to(dev)
        gm, gi = gx.reshape(-1, cin).mean(0), 1.0 / torch.sqrt(gx.reshape(-1, cin).var(0, unbiased=False) + 1e-5)
        if gate_kind == "affine":
            gate = ops.BnGateCtx(gx, gm, gi, scale=gi.clone(), shift=(-gm * gi))
        else:
            mask = torch.randint(0, 16, (m * cin // 4,), dtype=torch.uint8, generator=torch.Generator().manual_seed(8)).to(dev)
            gate = ops.BnGateCtx(gx, gm, gi, mask=mask)
    d1 = ops.conv2d_dgrad(dx, w, xin.shape, 1, 0, addend=addend.clone(), gate=gate)
    d2 = ops.conv2d_dgrad(lazy, w, xin.shape, 1, 0, addend=addend.clone(), gate=gate)
    close(d2, d1, rtol=2e-5, what="data gradient from the lazy dx")
    if gate is not None:
        for a, b in zip(d1._gate_partials[:2], d2._gate_partials[:2]):
            close(b.sum(0), a.sum(0), rtol=1e-4, what="gate partial sums")


@pytest.mark.parametrize("n,h,k,c,lazy_dy", [(3, 7, 64, 256, False), (2, 9, 128, 512, True), (5, 5, 64, 64, False)])
def test_gate_epilogue_also_reduces_against_the_projection_shortcut_input(dev, n, h, k, c, lazy_dy):
    """ssv_bn_gate's second target: the gated gradient g of a unit's closing activation is also the gradient w.r.t. the projection shortcut's
    BatchNorm output, so the stride-1 data gradient that produces g leaves sum g * xhat2 against THAT BatchNorm's input next to the usual
    (sum g, sum g * xhat) - with a plain and with a formed-on-load (LazyGrad) dY operand."""
    from ssv_amd import ops
    m = n * h * h
    w = (seeded_randn(951, k, c, 1, 1) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)      # conv1 of the next unit: c -> k
    dy = seeded_randn(952, n, h, h, k).to(dev)
    x1 = (seeded_randn(953, n, h, h, c) * 1.3 + 0.4).to(dev)          # input of the closing BatchNorm
    x2 = (seeded_randn(954, n, h, h, c) * 0.7 - 0.2).to(dev)          # input of the projection shortcut's BatchNorm
    stats = lambda t: (t.reshape(m, c).mean(0), 1.0 / torch.sqrt(t.reshape(m, c).var(0, unbiased=False) + 1e-5))
    (m1, i1), (m2, i2) = stats(x1), stats(x2)
    mask = torch.randint(0, 16, (m * c // 4,), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).to(dev)
    addend = seeded_randn(955, n, h, h, c).to(dev)
    gate = ops.BnGateCtx(x1, m1, i1, mask=mask, second=(x2, m2, i2))
    src = dy
    if lazy_dy:
        xb = seeded_randn(956, n, h, h, k).to(dev)
        coef = torch.stack([torch.rand(k, generator=torch.Generator().manual_seed(3)) + 0.5, torch.zeros(k), torch.zeros(k), torch.zeros(k)]).to(dev).contiguous()
        src = ops.LazyGrad(dy, xb, coef)                              # dx = A * g (B = D = 0): the plain product scaled per channel
        dy_eff = dy * coef[0]
    else:
        dy_eff = dy
    g = ops.conv2d_dgrad(src, w, (n, h, h, c), 1, 0, addend=addend.clone(), gate=gate)
    bits = torch.stack([(mask >> e) & 1 for e in range(4)], dim=1).reshape(m, c).bool()
    want = torch.where(bits, (dy_eff.reshape(m, k).double() @ w.reshape(k, c).double() + addend.reshape(m, c).double()), torch.zeros((), dtype=torch.float64, device=dev))
    close(g.reshape(m, c), want.float(), rtol=1e-4, what="gated gradient")
    pg, pgx, _ = g._gate_partials
    pg2, pgx2, _ = g._gate_partials_res
    assert pg2.data_ptr() == pg.data_ptr()
    xh1 = (x1.reshape(m, c).double() - m1.double()) * i1.double()
    xh2 = (x2.reshape(m, c).double() - m2.double()) * i2.double()
    close(pg.sum(0), want.sum(0).float(), rtol=1e-4, what="sum g")
    close(pgx.sum(0), (want * xh1).sum(0).float(), rtol=1e-4, what="sum g * xhat")
    close(pgx2.sum(0), (want * xh2).sum(0).float(), rtol=1e-4, what="sum g * xhat2")


@pytest.mark.parametrize("n,h,c,k", [(5, 9, 64, 64), (3, 12, 256, 64), (2, 14, 128, 256), (3, 7, 96, 132)])
@pytest.mark.parametrize("res_affine", [False, True])
def test_conv_that_forms_and_writes_the_closing_activation_is_bitwise_bn_apply_then_conv(dev, n, h, c, k, res_affine):
    """ssv_conv2d_fwd_sumin_stats: a = relu(x * scale + shift + (res | res * rscale + rshift)) formed while conv1 stages its input and written
    by the same kernel.  Same fmaf / add / fmaxf on the same floats as ssv_bn_apply, the same GEMM on that operand as
    ssv_conv2d_fwd_stats: the activation, its ReLU byte mask, the convolution output and the statistics partials are IDENTICAL bits -
    identity shortcut and projection shortcut with its own BatchNorm affine, narrow (256 x 64) and wide (128 x 128) tiles, ragged sizes."""
    from ssv_amd import ops
    x, res = seeded_randn(31, n, h, h, c).to(dev), seeded_randn(32, n, h, h, c).to(dev)
    w = (seeded_randn(33, k, c, 1, 1) * 0.1).contiguous(memory_format=torch.channels_last).to(dev)
    scale, shift = (seeded_randn(34, c) * 0.3 + 1.0).to(dev), (seeded_randn(35, c) * 0.2).to(dev)
    raff = ((seeded_randn(36, c) * 0.3 + 1.0).to(dev), (seeded_randn(37, c) * 0.2).to(dev)) if res_affine else None
    a_ref, m_ref = ops.bn_apply(x, scale, shift, relu=True, residual=res, res_affine=raff, want_mask=True)
    y_ref, pm_ref, p2_ref = ops.conv2d_fwd_stats(a_ref, w, 1, 0)
    y, (pm, p2), a, mask = ops.conv2d_fwd_sumin(x, res, scale, shift, raff, w, want_mask=True)
    assert torch.equal(a, a_ref) and torch.equal(mask, m_ref)
    assert torch.equal(y, y_ref) and torch.equal(pm, pm_ref) and torch.equal(p2, p2_ref)
    y2, _, a2, mask2 = ops.conv2d_fwd_sumin(x, res, scale, shift, raff, w, want_mask=False)
    assert mask2 is None and torch.equal(a2, a_ref) and torch.equal(y2, y_ref)
    # and against an fp64 evaluation (the bitwise statements above are between two of our own kernels)
    ar = torch.relu(x.double() * scale.double() + shift.double() + (res.double() if raff is None else res.double() * raff[0].double() + raff[1].double()))
    np.testing.assert_allclose(a.cpu().double().numpy(), ar.cpu().numpy(), rtol=1e-5, atol=1e-5)
    yr = torch.einsum("nhwc,kc->nhwk", ar, w.double().view(k, c))
    np.testing.assert_allclose(y.cpu().double().numpy(), yr.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,h,w_,k,r,stride,pad", [(3, 20, 20, 64, 7, 2, 3), (2, 9, 11, 64, 3, 1, 1), (2, 13, 7, 132, 7, 2, 3), (1, 7, 7, 16, 3, 1, 1), (5, 33, 31, 64, 7, 2, 3),
                                                    (2, 224, 224, 64, 7, 2, 3), (3, 32, 32, 64, 3, 1, 1), (2, 40, 24, 64, 7, 2, 3), (9, 12, 8, 64, 5, 2, 2), (1, 8, 228, 64, 7, 2, 3)])
def test_stem_conv_in_the_row_taps_form_on_the_unpadded_image(dev, n, h, w_, k, r, stride, pad):
    """ssv_stem_conv_fwd / ssv_stem_conv_wgrad: the 3-channel stem without channel padding - a k-tile is one filter row, 3 S contiguous floats of
    the image row at a 12-byte-aligned address, left / right image borders masked per element - against torch fp64, with the statistics epilogue;
    odd sizes, non-square images, ragged channel tiles (K = 132)."""
    from ssv_amd import ops
    x3 = seeded_randn(1, n, h, w_, 3)
    w3 = seeded_randn(2, k, 3, r, r) * 0.2
    xr, wr = x3.permute(0, 3, 1, 2).double(), w3.double().requires_grad_()
    ref = F.conv2d(xr, wr, stride=stride, padding=pad)
    dy = seeded_randn(3, *ref.permute(0, 2, 3, 1).shape)
    ref.backward(dy.permute(0, 3, 1, 2).double())
    wd = w3.contiguous(memory_format=torch.channels_last).to(dev)
    assert ops.can_row_stem(tuple(wd.shape))
    wrows = ops.stem_weight_rows(wd)
    assert wrows.shape == (k * r, 24) and float(wrows[:, 3 * r:].abs().max()) == 0.0
    y, part = ops.stem_conv_fwd(x3.to(dev), wrows, tuple(wd.shape), stride, pad, want_stats=True)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=2e-5)
    y0, none = ops.stem_conv_fwd(x3.to(dev), wrows, tuple(wd.shape), stride, pad, want_stats=False)
    assert none is None and torch.equal(y0, y)
    m = y.numel() // k
    y2 = y.view(m, k).cpu().double()
    rpg = ops._rows_per_group(part)              # 64 output pixels per partial, or whole output rows on the rows-in-LDS kernel
    assert part[0].shape == ((m + rpg - 1) // rpg, k) and (rpg == 64 or m % rpg == 0)
    for g in range(0, part[0].shape[0], max(1, part[0].shape[0] // 40)):
        blk = y2[rpg * g:rpg * g + rpg]
        np.testing.assert_allclose(part[0][g].cpu().numpy(), blk.mean(0).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(part[1][g].cpu().numpy(), ((blk - blk.mean(0)) ** 2).sum(0).numpy(), rtol=1e-3, atol=1e-4)
    # ... and BatchNorm from the partials == BatchNorm from its own statistics pass
    gamma, beta = (seeded_randn(7, k).abs() + 0.5).to(dev), seeded_randn(8, k).to(dev)
    rm, rv, nbt = torch.zeros(k, device=dev), torch.ones(k, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    mean, invstd, _, _ = ops.bn_stats_finalize(m, k, part, gamma, beta, rm, rv, nbt)
    np.testing.assert_allclose(mean.cpu().double().numpy(), y2.mean(0).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(invstd.cpu().double().numpy(), (1.0 / torch.sqrt(y2.var(0, unbiased=False) + 1e-5)).numpy(), rtol=1e-4)
    dwr = ops.stem_conv_wgrad(x3.to(dev), dy.to(dev), tuple(wd.shape), stride, pad)
    assert dwr.shape == (k * r, 24) and float(dwr[:, 3 * r:].abs().max()) == 0.0            # the padding columns see zero operands
    got = dwr[:, :3 * r].reshape(k, r, r, 3).permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.cpu().double().numpy(), wr.grad.numpy(), rtol=2e-4, atol=2e-4)
