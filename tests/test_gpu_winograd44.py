"""GPU: Winograd F(4x4, 3x3) (csrc/winograd44.hip + 36 batched implicit-GEMM launches) for the forward and the data gradient of the deep stride-1 3x3 layers
(networks/resnet.py:7-10, 56-58) against torch fp64 convolutions: forward with the fused input BatchNorm and the statistics epilogue (one partial per row of tiles or per
image), the F(2x2) operand it leaves for the weight gradient (bitwise the one F(2x2)'s own transform writes), data gradient with the ReLU gate (recomputed and byte-mask)
and its partial sums, ragged maps, and the per-product dispatch rule.  Error bounds: F(4x4)'s transforms multiply by constants up to 8 and 16/15, so it is 2-3x further
from fp64 than the direct kernels (profiles/r04_probe_winograd44.txt) - the bound here is 4e-6 where F(2x2)'s is 2e-6."""
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


def rel(a, b):
    return float((a.detach().cpu().double() - b).norm() / (b.norm() + 1e-30))


# (n, h, w, c, k): the three ResNet-50 shapes, ragged maps in both directions, one tile, maps smaller than a tile, wide / narrow channel counts
SHAPES = [(6, 14, 14, 256, 256), (5, 7, 7, 512, 512), (3, 28, 28, 128, 128), (4, 8, 6, 128, 256), (3, 9, 5, 256, 128), (2, 7, 8, 1024, 128), (2, 4, 4, 128, 2048), (3, 3, 2, 128, 128)]


def _case(n, h, w_, c, k, seed=0):
    x = seeded_randn(seed + 1, n, h, w_, c)
    w = seeded_randn(seed + 2, k, c, 3, 3) * (2.0 / (9 * c)) ** 0.5
    dy = seeded_randn(seed + 3, n, h, w_, k)
    aff = (seeded_randn(seed + 4, c) * 0.3 + 1.0, seeded_randn(seed + 5, c) * 0.2)
    return x, w, dy, aff


@pytest.mark.parametrize("n,h,w_,c,k", SHAPES)
@pytest.mark.parametrize("affine", [False, True])
def test_forward_statistics_and_the_f22_operand_it_leaves(dev, n, h, w_, c, k, affine):
    from ssv_amd import _lib, ops
    lib = _lib.load()
    x, w, _, aff = _case(n, h, w_, c, k)
    a = torch.relu(x.double() * aff[0].double() + aff[1].double()) if affine else x.double()
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    xd, wd = x.to(dev), w.contiguous(memory_format=torch.channels_last).to(dev)
    affd = (aff[0].to(dev), aff[1].to(dev)) if affine else None
    prev44, ops.WINOGRAD44_WGRAD = ops.WINOGRAD44_WGRAD, False            # round 4's selection: the forward leaves the F(2x2) operand for the weight gradient
    try:
        y, part, v2 = ops.wino44_conv2d_fwd(xd, wd, in_affine=affd, want_stats=True, keep_v=True)
    finally:
        ops.WINOGRAD44_WGRAD = prev44
    assert rel(y, ref) < (4e-6 if max(c, k) <= 512 else 8e-6), rel(y, ref)        # beyond 512 channels (not dispatched: ops.WINOGRAD44_MAX_CHANNELS) the sums are longer
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), rtol=1e-4, atol=4e-5 * float(ref.abs().max()))
    # statistics partials: one per row of tiles (4 x W pixels) when H % 4 == 0, else one per image - equal groups either way
    rpg = 4 * w_ if h % 4 == 0 else h * w_
    assert part[2] == rpg == int(lib.ssv_wino44_stats_rows_per_group(n, h, w_)) and part[0].shape == (n * h * w_ // rpg, k)
    gamma, beta = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    rm, rv, nbt = torch.zeros(k, device=dev), torch.ones(k, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    mean, invstd, _, _ = ops.bn_stats_finalize(n * h * w_, k, part, gamma, beta, rm, rv, nbt)
    r2 = ref.reshape(-1, k)
    np.testing.assert_allclose(mean.cpu().double().numpy(), r2.mean(0).numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(invstd.cpu().double().numpy(), (1.0 / torch.sqrt(r2.var(0, unbiased=False) + 1e-5)).numpy(), rtol=1e-4)
    # the F(2x2) transformed input written beside F(4x4)'s own: the bits of F(2x2)'s own input transform
    prev, ops.WINOGRAD44 = ops.WINOGRAD44, False
    try:
        _, _, v2_ref = ops.wino_conv2d_fwd(xd, wd, in_affine=affd, want_stats=False, keep_v=True)
    finally:
        ops.WINOGRAD44 = prev
    assert tuple(v2.shape) == (16, n * ((h + 1) // 2) * ((w_ + 1) // 2), c) and torch.equal(v2, v2_ref)
    # ... so the weight gradient from it is F(2x2)'s, to its tolerance
    wr = w.double().clone().requires_grad_()
    dy = seeded_randn(77, n, h, w_, k)
    F.conv2d(a.permute(0, 3, 1, 2), wr, padding=1).backward(dy.double().permute(0, 3, 1, 2))
    dw = torch.zeros_like(wd)
    ops.wino_conv2d_wgrad(v2, dy.to(dev), wd, dw, accumulate=False)
    assert rel(dw, wr.grad) < 2e-6
    # without the statistics and without the kept operand: the same y
    y2, none, nov = ops.wino44_conv2d_fwd(xd, wd, in_affine=affd, want_stats=False, keep_v=False)
    assert none is None and nov is None and torch.equal(y2, y)
    # the shipped selection (ops.WINOGRAD44_WGRAD): the kept operand is F(4x4)'s own transformed input, the weight gradient runs through F(4x4) on it
    assert ops.WINOGRAD44_WGRAD
    y3, _, v4 = ops.wino44_conv2d_fwd(xd, wd, in_affine=affd, want_stats=False, keep_v=True)
    assert torch.equal(y3, y) and tuple(v4.shape) == (36, n * ((h + 3) // 4) * ((w_ + 3) // 4), c)
    dw4 = torch.zeros_like(wd)
    ops.wino_conv2d_wgrad(v4, dy.to(dev), wd, dw4, accumulate=False)                 # told apart by the leading dimension
    assert rel(dw4, wr.grad) < (4e-6 if max(c, k) <= 512 else 8e-6), rel(dw4, wr.grad)
    ops.wino_conv2d_wgrad(v4, dy.to(dev), wd, dw4, accumulate=True)
    assert rel(dw4, 2 * wr.grad) < (4e-6 if max(c, k) <= 512 else 8e-6)
    with pytest.raises(_lib.SsvError):
        ops.wino44_conv2d_wgrad(v4[:, :-1], dy.to(dev), wd, dw4)


def _wgrad64(a, dy):
    """fp64 weight gradient [K][C][3][3] of a 3x3 / padding 1 convolution from NHWC device operands: nine fp64 GEMMs (test infrastructure)."""
    n, h, w_, c = a.shape
    k = dy.shape[3]
    ap = torch.zeros((n, h + 2, w_ + 2, c), dtype=torch.float64, device=a.device)
    ap[:, 1:-1, 1:-1] = a.double()
    d2 = dy.double().reshape(-1, k)
    out = torch.empty((k, c, 3, 3), dtype=torch.float64, device=a.device)
    for r in range(3):
        for s in range(3):
            out[:, :, r, s] = d2.t() @ ap[:, r:r + h, s:s + w_].reshape(-1, c)
    return out


@pytest.mark.parametrize("n,h,c", [(37, 28, 128), (24, 14, 256), (9, 56, 64), (130, 7, 512), (3, 12, 128)])
def test_weight_gradient_through_f44_on_ragged_tile_counts(dev, n, h, c):
    """The blocked accumulation of the F(4x4) weight gradient (row chunks of 512 tiles folded in fp64, a flush every 128 rows inside the kernel) on tile counts that are
    no multiple of either: 37 x 49 = 1,813 (three chunks + 277 = two flushes + 21), 24 x 16 = 384 (below one chunk), 9 x 196 = 1,764 on the 64-channel layer, 130 x 4 = 520
    (one chunk + 8), 3 x 9 = 27 (below one flush; 12 x 12 maps) - same bar as at the bench batch."""
    from ssv_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n, h, h, c, device=dev, generator=g)
    w = (torch.randn(c, c, 3, 3, device=dev, generator=g) * (2.0 / (9 * c)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, h, h, c, device=dev, generator=g)
    aff = (torch.rand(c, device=dev, generator=g) + 0.5, torch.randn(c, device=dev, generator=g) * 0.1)
    ref = _wgrad64(torch.relu(x * aff[0] + aff[1]), dy)
    _, _, v4 = ops.wino44_conv2d_fwd(x, w, in_affine=aff, want_stats=False, keep_v=True)
    assert v4.shape[0] == 36
    dw = torch.full_like(w, 7.0)                               # accumulate=False must overwrite
    ops.wino_conv2d_wgrad(v4, dy, w, dw, accumulate=False)
    err = float((dw.double() - ref).norm() / ref.norm())
    assert err <= 2e-6, err
    acc = dw.clone()
    ops.wino_conv2d_wgrad(v4, dy, w, acc, accumulate=True)     # ... and accumulate=True adds exactly one more copy
    err2 = float((acc.double() - 2 * ref).norm() / (2 * ref.norm()))
    assert err2 <= 2e-6, err2


@pytest.mark.parametrize("h,c", [(28, 128), (14, 256), (7, 512)])
def test_weight_gradient_through_f44_at_batch_512_within_2e6_of_fp64(dev, h, c):
    """The bar the F(4x4) weight gradient ships under (round-4 review): relative l2 error of dW against fp64 <= 2e-6 on the three ResNet-50 Winograd shapes AT THE
    BENCH BATCH (the transformed-domain sums run over 25,088 / 8,192 / 2,048 tiles), the tolerance the F(2x2) weight gradient is held to above.  Operand: a
    BatchNorm + ReLU output (non-negative, as in the network), formed on load by the forward's input transform."""
    from ssv_amd import ops
    n = 512
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n, h, h, c, device=dev, generator=g)
    w = (torch.randn(c, c, 3, 3, device=dev, generator=g) * (2.0 / (9 * c)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, h, h, c, device=dev, generator=g)
    aff = (torch.rand(c, device=dev, generator=g) + 0.5, torch.randn(c, device=dev, generator=g) * 0.1)
    ref = _wgrad64(torch.relu(x * aff[0] + aff[1]), dy)
    _, _, v4 = ops.wino44_conv2d_fwd(x, w, in_affine=aff, want_stats=False, keep_v=True)
    assert v4.shape[0] == 36
    dw = torch.zeros_like(w)
    ops.wino_conv2d_wgrad(v4, dy, w, dw, accumulate=False)
    err = float((dw.double() - ref).norm() / ref.norm())
    assert err <= 2e-6, err


@pytest.mark.parametrize("n,h,w_,c,k", SHAPES)
def test_data_gradient_with_the_relu_gate_epilogue(dev, n, h, w_, c, k):
    from ssv_amd import ops
    x, w, dy, aff = _case(n, h, w_, c, k, seed=10)
    refdx = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    wd, dyd = w.contiguous(memory_format=torch.channels_last).to(dev), dy.to(dev)
    dx = ops.wino44_conv2d_dgrad(dyd, wd)
    assert rel(dx, refdx) < (4e-6 if max(c, k) <= 512 else 8e-6), rel(dx, refdx)
    gx = seeded_randn(20, n, h, w_, c)
    mean, invstd = seeded_randn(21, c) * 0.1, seeded_randn(22, c).abs() + 0.5
    bit = (gx * aff[0] + aff[1]) > 0
    gate = ops.BnGateCtx(gx.to(dev), mean.to(dev), invstd.to(dev), scale=aff[0].to(dev), shift=aff[1].to(dev))
    g = ops.wino44_conv2d_dgrad(dyd, wd, gate=gate)
    want = torch.where(bit, refdx, torch.zeros_like(refdx))
    near = ((gx * aff[0] + aff[1]).abs() < 1e-6)
    assert int(near.sum()) < 4
    diff = (g.cpu().double() - want).abs()
    diff[near] = 0
    assert float(diff.max()) < 8e-5 * float(refdx.abs().max())
    bits = bit.reshape(-1, 4).to(torch.uint8)
    mask = (bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3)).to(torch.uint8)
    g2 = ops.wino44_conv2d_dgrad(dyd, wd, gate=ops.BnGateCtx(gx.to(dev), mean.to(dev), invstd.to(dev), mask=mask.to(dev)))
    if int(near.sum()) == 0:
        assert torch.equal(g2, g) and torch.equal(g2._gate_partials[0], g._gate_partials[0]) and torch.equal(g2._gate_partials[1], g._gate_partials[1])
    sg, sgx, groups = g._gate_partials
    assert groups == sg.shape[0] == n * ((h + 3) // 4)                 # one partial per row of tiles
    xhat = (gx.double() - mean.double()) * invstd.double()
    np.testing.assert_allclose(sg.sum(0).cpu().double().numpy(), want.reshape(-1, c).sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(want.abs().sum(dim=(0, 1, 2)).max()))
    np.testing.assert_allclose(sgx.sum(0).cpu().double().numpy(), (want * xhat).reshape(-1, c).sum(0).numpy(), rtol=1e-4, atol=1e-4 * float((want * xhat).abs().sum(dim=(0, 1, 2)).max()))


def test_which_product_takes_f44(dev):
    """Per product, by the transformed-domain work F(4x4) leaves (36 x tiles of 4 against 16 x tiles of 2): 28x28 and 7x7 (0.5625) switch forward and data gradient;
    14x14 (0.735: F(2x2) tiles it exactly) the data gradient and - since the weight gradient runs F(4x4) on the forward's own transformed input (ops.WINOGRAD44_WGRAD,
    no second operand to write) - the forward as well; with round 4's selection (WINOGRAD44_WGRAD off) its forward stays on F(2x2).
    SSV_WINOGRAD44=0 / ops.WINOGRAD44 = False restores F(2x2) everywhere."""
    from ssv_amd import ops
    assert ops.WINOGRAD44 and ops.WINOGRAD44_WGRAD
    assert abs(ops._wino44_ratio(28, 28) - 0.5625) < 1e-9 and abs(ops._wino44_ratio(7, 7) - 0.5625) < 1e-9 and abs(ops._wino44_ratio(14, 14) - 36 * 16 / (16 * 49)) < 1e-9
    calls = {"f": 0, "d": 0}
    inner_f, inner_d = ops.wino44_conv2d_fwd, ops.wino44_conv2d_dgrad

    def cf(*a, **k):
        calls["f"] += 1
        return inner_f(*a, **k)

    def cd(*a, **k):
        calls["d"] += 1
        return inner_d(*a, **k)
    ops.wino44_conv2d_fwd, ops.wino44_conv2d_dgrad = cf, cd
    try:
        for hw, ch, f44, d44, f44_r4 in ((28, 128, 1, 1, 1), (14, 256, 1, 1, 0), (7, 512, 1, 1, 1)):
            x = seeded_randn(5, 1024 // ((hw + 3) // 4) ** 2 + 1, hw, hw, ch).to(dev)         # just past ops.WINOGRAD44_MIN_TILES tiles
            w = (seeded_randn(6, ch, ch, 3, 3) * 0.02).contiguous(memory_format=torch.channels_last).to(dev)
            calls.update(f=0, d=0)
            v = ops.wino_conv2d_fwd(x, w, keep_v=True)[2]
            ops.wino_conv2d_dgrad(x, w)
            assert (calls["f"], calls["d"]) == (f44, d44) and v.shape[0] == (36 if f44 else 16), (hw, calls)
            ops.WINOGRAD44_WGRAD = False
            try:
                calls.update(f=0, d=0)
                v = ops.wino_conv2d_fwd(x, w, keep_v=True)[2]
                assert calls["f"] == f44_r4 and v.shape[0] == 16, (hw, calls)
            finally:
                ops.WINOGRAD44_WGRAD = True
        prev, ops.WINOGRAD44 = ops.WINOGRAD44, False
        try:
            calls.update(f=0, d=0)
            ops.wino_conv2d_fwd(x, w)
            ops.wino_conv2d_dgrad(x, w)
            assert calls == {"f": 0, "d": 0}
        finally:
            ops.WINOGRAD44 = prev
        # too few tiles (small batches): F(2x2)
        calls.update(f=0, d=0)
        ops.wino_conv2d_fwd(x[:8], w)
        ops.wino_conv2d_dgrad(x[:8], w)
        assert calls == {"f": 0, "d": 0}
        # beyond the channel count the numerics bar was measured for: F(2x2)
        x = seeded_randn(7, 260, 8, 8, 1024).to(dev)
        w = (seeded_randn(8, 1024, 1024, 3, 3) * 0.01).contiguous(memory_format=torch.channels_last).to(dev)
        calls.update(f=0, d=0)
        ops.wino_conv2d_fwd(x, w)
        ops.wino_conv2d_dgrad(x, w)
        assert calls == {"f": 0, "d": 0}
    finally:
        ops.wino44_conv2d_fwd, ops.wino44_conv2d_dgrad = inner_f, inner_d


@pytest.mark.parametrize("n,h,w_,c,k", SHAPES[:6])
def test_one_pass_over_dy_leaves_both_backward_operands(dev, n, h, w_, c, k):
    """ssv_wino44_dy_transform_both: the data gradient's transformed input and the weight gradient's transformed dY from ONE pass over dy - bit for bit what the two
    separate transforms write; with the formed-on-load operand (g, x, coefficients of the BatchNorm backward) equal to the transforms of the materialised dy."""
    from ssv_amd import _lib, ops
    lib = _lib.load()
    dy = seeded_randn(41, n, h, w_, k).to(dev)
    t = int(lib.ssv_wino44_tiles(n, h, w_))
    vd, dm = torch.empty((36, t, k), device=dev), torch.empty((36, t, k), device=dev)
    _lib.call("ssv_wino44_dy_transform_both", n, h, w_, k, _lib.ptr(dy), None, _lib.ptr(vd), _lib.ptr(dm), _lib.stream())
    vd_ref, dm_ref = torch.empty_like(vd), torch.empty_like(dm)
    _lib.call("ssv_wino44_input_transform", n, h, w_, k, _lib.ptr(dy), None, None, _lib.ptr(vd_ref), None, _lib.stream())
    _lib.call("ssv_wino44_dy_transform", n, h, w_, k, _lib.ptr(dy), _lib.ptr(dm_ref), _lib.stream())
    assert torch.equal(vd, vd_ref) and torch.equal(dm, dm_ref)
    # formed on load: dy = A g + B (x - mean) + D per channel
    g, x = seeded_randn(42, n, h, w_, k), seeded_randn(43, n, h, w_, k)
    coef = torch.stack([seeded_randn(44, k) * 0.5 + 1.0, seeded_randn(45, k) * 0.1, seeded_randn(46, k) * 0.3, seeded_randn(47, k) * 0.05])
    want = (coef[0].double() * g.double() + coef[2].double() * (x.double() - coef[1].double()) + coef[3].double()).float().to(dev).contiguous()
    _lib.call("ssv_wino44_input_transform", n, h, w_, k, _lib.ptr(want), None, None, _lib.ptr(vd_ref), None, _lib.stream())
    _lib.call("ssv_wino44_dy_transform", n, h, w_, k, _lib.ptr(want), _lib.ptr(dm_ref), _lib.stream())
    gd, xd, cd = g.to(dev), x.to(dev), coef.to(dev).contiguous()
    dyin = _lib.BnDyin(_lib.ptr(xd), _lib.ptr(cd))
    import ctypes
    _lib.call("ssv_wino44_dy_transform_both", n, h, w_, k, _lib.ptr(gd), ctypes.byref(dyin), _lib.ptr(vd), _lib.ptr(dm), _lib.stream())
    for got, ref in ((vd, vd_ref), (dm, dm_ref)):
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), float((got - ref).abs().max() / ref.abs().max())
    # ... and through the host path: the weight gradient leaves the data gradient's operand on the LazyGrad, the data gradient picks it up
    w = (seeded_randn(48, k, c, 3, 3) * (2.0 / (9 * c)) ** 0.5).contiguous(memory_format=torch.channels_last).to(dev)
    xin = seeded_randn(49, n, h, w_, c).to(dev)
    _, _, v4 = ops.wino44_conv2d_fwd(xin, w, want_stats=False, keep_v=True)
    lazy = ops.LazyGrad(gd, xd, cd)
    dw = torch.zeros_like(w)
    ops.conv2d_wgrad(xin, lazy, w, dw, 1, 1, accumulate=False, wino_v=v4)
    assert lazy.wino_vd is not None
    dx = ops.conv2d_dgrad(lazy, w, xin.shape, 1, 1)
    assert lazy.wino_vd is None
    dw_ref = torch.zeros_like(w)
    ops.wino44_conv2d_wgrad(v4, want, w, dw_ref, accumulate=False)
    dx_ref = ops.wino44_conv2d_dgrad(want, w)
    assert rel(dw, dw_ref.cpu().double()) < 2e-5 and rel(dx, dx_ref.cpu().double()) < 2e-5
    with pytest.raises(_lib.SsvError):                      # a LazyGrad that the weight gradient has not visited cannot feed the Winograd data gradient
        ops.wino44_conv2d_dgrad(ops.LazyGrad(gd, xd, cd), w)
