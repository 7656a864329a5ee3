"""The SSV_ARITH_BF16X3 arithmetic (csrc/split_bf16.h): every fp32 product as six exact bf16 piece products, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
What is asserted here, on the GPU, through the C ABI (ops.* -> ssv_conv2d_* / ssv_gemm_batched_*):

  * on ALL 53 convolutions of networks/resnet.py's ResNet-50 at 224 x 224 x their three products (forward, data gradient, weight gradient) the error against an
    fp64 evaluation is <= 1.05 x the fp32-MFMA kernel's on the same operands (the reviewer's bar for making this arithmetic the default);
  * the same for the batched transformed-domain products of the Winograd layers, ragged shapes and the bias / addend epilogue;
  * what happens at the edges of fp32's range: +-0, 1e-38, 1e+38, Inf, NaN (decided in csrc/split_bf16.h, pinned here).
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from ssv_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _r50_convs():
    """(name, C, K, R, stride, pad, H_in) of the 53 convolutions of resnet50() with the 7x7 stem at 224 x 224 (networks/resnet.py:80-144)."""
    convs = [("conv1", 3, 64, 7, 2, 3, 224)]
    inplanes, h = 64, 56
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
        for b in range(blocks):
            st = stride if b == 0 else 1
            convs.append((f"layer{li}.{b}.conv1", inplanes, planes, 1, 1, 0, h))
            convs.append((f"layer{li}.{b}.conv2", planes, planes, 3, st, 1, h))          # the stride sits on the 3x3 (networks/resnet.py:58)
            convs.append((f"layer{li}.{b}.conv3", planes, planes * 4, 1, 1, 0, h // st))
            if b == 0:
                convs.append((f"layer{li}.{b}.downsample", inplanes, planes * 4, 1, st, 0, h))
            h //= st
            inplanes = planes * 4
    assert len(convs) == 53
    return convs


R50 = _r50_convs()
SHAPES = {}
for _name, *_shape in R50:
    SHAPES.setdefault(tuple(_shape), []).append(_name)


def _rel(got, ref):
    return float((got.double() - ref).norm() / ref.norm())


@pytest.mark.parametrize("shape", sorted(SHAPES), ids=lambda s: "C%d_K%d_R%d_s%d_p%d_H%d" % s)
def test_all_53_layer_shapes_x_3_products_no_worse_than_fp32_mfma(dev, shape):
    """Every distinct (C, K, R, stride, pad, H) among the 53 convolutions (SHAPES maps each to the layers that have it), batch 8: the three products in both
    arithmetics against fp64.  The stem (C = 3) has no bf16-piece kernel: ssv_conv_arithmetic says so and its two arithmetics are the same launch."""
    from ssv_amd import _lib, ops
    c, k, r, stride, pad, h = shape
    n = 8
    g = torch.Generator(device=dev).manual_seed(hash(shape) % (1 << 31))
    x = torch.relu(torch.randn(n, h, h, c, device=dev, generator=g) + 0.3)                                   # a ReLU output, as every conv input but the image is
    w = (torch.randn(k, c, r, r, device=dev, generator=g) * (2.0 / (c * r * r)) ** 0.5).contiguous(memory_format=torch.channels_last)
    ho = (h + 2 * pad - r) // stride + 1
    dy = torch.randn(n, ho, ho, k, device=dev, generator=g)
    xd, wd, dyd = x.permute(0, 3, 1, 2).double(), w.double().contiguous(), dy.permute(0, 3, 1, 2).double()
    ref_y = F.conv2d(xd, wd, stride=stride, padding=pad).permute(0, 2, 3, 1)
    ref_dx = torch.nn.grad.conv2d_input(xd.shape, wd, dyd, stride=stride, padding=pad).permute(0, 2, 3, 1)
    ref_dw = torch.nn.grad.conv2d_weight(xd, wd.shape, dyd, stride=stride, padding=pad)
    err = {}
    for arith in ("f32", "bf16x3"):
        with ops.arithmetic(arith):
            y = ops.conv2d_fwd(x, w, stride, pad)
            if c == 3:
                dx = None
            else:
                dx = ops.conv2d_dgrad(dy, w, (n, h, h, c), stride, pad)
            dw = torch.zeros_like(w)
            ops.conv2d_wgrad(x, dy, w, dw, stride, pad, accumulate=False)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and torch.isfinite(dw).all()
            err[arith] = (_rel(y, ref_y), None if dx is None else _rel(dx, ref_dx), _rel(dw, ref_dw))
    d = ops.conv_desc((n, h, h, c), (k, c, r, r), stride, pad, w=w)
    lib = _lib.load()
    with ops.arithmetic("bf16x3"):
        d = ops.conv_desc((n, h, h, c), (k, c, r, r), stride, pad, w=w)
        used_fwd, used_wgrad = lib.ssv_conv_arithmetic(C.byref(d), 0), lib.ssv_conv_arithmetic(C.byref(d), 2)
    if c % 32 == 0:
        assert used_fwd == _lib.ARITH_BF16X3 and used_wgrad == _lib.ARITH_BF16X3, "every layer but the stem has a bf16-piece kernel for all three products"
    else:
        assert used_fwd == _lib.ARITH_F32_MFMA
    for i, what in enumerate(("forward", "data gradient", "weight gradient")):
        e32, esp = err["f32"][i], err["bf16x3"][i]
        if e32 is None:
            continue
        assert esp <= 1.05 * e32 + 1e-9, f"{what} of {SHAPES[shape][0]} (x{len(SHAPES[shape])}): bf16x3 {esp:.3e} vs fp32 MFMA {e32:.3e} against fp64"
        assert esp < 2e-6, (what, esp)


@pytest.mark.parametrize("nb,t,c,k", [(36, 1813, 128, 128), (16, 520, 256, 256), (36, 100, 512, 512), (4, 333, 64, 192), (2, 129, 1024, 132), (3, 700, 96, 64)])
def test_batched_products_ragged_shapes_bias_and_addend(dev, nb, t, c, k):
    """ssv_gemm_batched_split / ssv_gemm_batched_wgrad_split against fp64 and against the fp32-MFMA kernels: ragged row counts, channel counts that are no multiple of a
    tile (192, 132), a 64-wide product (the 256 x 64 tile), contractions of 64 ... 1,024; then the plain 1x1 / Linear epilogue (+ bias + addend, in place)."""
    from ssv_amd import _lib, ops
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn(nb, t, c, device=dev, generator=g).clamp_min_(-0.5)
    w = torch.randn(nb, k, c, device=dev, generator=g) * (1.0 / c) ** 0.5
    ref = torch.bmm(a.double(), w.double().transpose(1, 2))
    y32 = torch.full((nb, t, k), float("nan"), device=dev)
    ysp = torch.full((nb, t, k), float("nan"), device=dev)
    with ops.arithmetic("f32"):
        ops._gemm_batched(nb, t, c, k, a, w, y32)
    with ops.arithmetic("bf16x3"):
        ops._gemm_batched(nb, t, c, k, a, w, ysp)
    torch.cuda.synchronize()
    assert torch.isfinite(ysp).all()
    e32, esp = _rel(y32, ref), _rel(ysp, ref)
    assert esp <= 1.05 * e32 + 1e-9, (esp, e32)
    assert float((ysp.double() - ref).abs().max()) <= 1.5 * float((y32.double() - ref).abs().max()) + 1e-12
    # the weight-gradient-shaped product dU[b] = dM[b]^T V[b]
    dm = torch.randn(nb, t, k, device=dev, generator=g)
    ref_du = torch.bmm(dm.double().transpose(1, 2), a.double())
    du32, dusp = torch.empty(nb, k, c, device=dev), torch.empty(nb, k, c, device=dev)
    with ops.arithmetic("f32"):
        ops._gemm_batched_wgrad(nb, t, c, k, a, dm, du32)
    with ops.arithmetic("bf16x3"):
        ops._gemm_batched_wgrad(nb, t, c, k, a, dm, dusp)
    torch.cuda.synchronize()
    assert _rel(dusp, ref_du) <= 1.05 * _rel(du32, ref_du) + 1e-9, (_rel(dusp, ref_du), _rel(du32, ref_du))
    # bias + addend (one product), the addend in place
    bias = torch.randn(k, device=dev, generator=g)
    acc = torch.randn(t, k, device=dev, generator=g)
    want = ref[0] + bias.double() + acc.double()
    with ops.arithmetic("bf16x3"):
        pl = ops._planes(w[0].contiguous())
    _lib.call("ssv_gemm_batched_split", 1, t, c, k, _lib.ptr(a[0]), _lib.ptr(pl), _lib.ptr(acc), _lib.ptr(bias), _lib.ptr(acc), _lib.stream())
    torch.cuda.synchronize()
    assert _rel(acc, want) <= 1.10 * e32 + 1e-7


@pytest.mark.parametrize("n,h,c", [(64, 28, 128), (64, 14, 256), (128, 7, 512), (16, 56, 64)])
def test_winograd_layers_keep_their_error_in_the_bf16x3_arithmetic(dev, n, h, c):
    """The F(4x4) / F(2x2) forward, data gradient and weight gradient as ops dispatches them: against fp64 no worse than with fp32-MFMA products (the weight gradient
    WITHOUT the register-level flush the fp32 form needs: the bf16 instruction folds 32 products per accumulator rounding)."""
    from ssv_amd import ops
    g = torch.Generator(device=dev).manual_seed(13)
    x = torch.relu(torch.randn(n, h, h, c, device=dev, generator=g))
    w = (torch.randn(c, c, 3, 3, device=dev, generator=g) * (2.0 / (9 * c)) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, h, h, c, device=dev, generator=g)
    xd, wd, dyd = x.permute(0, 3, 1, 2).double(), w.double().contiguous(), dy.permute(0, 3, 1, 2).double()
    ref_y = F.conv2d(xd, wd, padding=1).permute(0, 2, 3, 1)
    ref_dx = F.conv_transpose2d(dyd, wd, padding=1).permute(0, 2, 3, 1)
    ref_dw = torch.nn.grad.conv2d_weight(xd, wd.shape, dyd, padding=1)
    errs = {}
    for arith in ("f32", "bf16x3"):
        with ops.arithmetic(arith):
            y, _, v = ops.wino44_conv2d_fwd(x, w, want_stats=False, keep_v=True)
            dx = ops.wino44_conv2d_dgrad(dy.clone(), w)
            dw = torch.zeros_like(w)
            ops.wino_conv2d_wgrad(v, dy.clone(), w, dw, accumulate=False, dgrad_follows=False)
            torch.cuda.synchronize()
            errs[arith] = (_rel(y, ref_y), _rel(dx, ref_dx), _rel(dw, ref_dw))
    for i in range(3):
        assert errs["bf16x3"][i] <= 1.05 * errs["f32"][i] + 1e-9, errs
    assert errs["bf16x3"][2] < 2e-6, errs          # the weight gradient's bar (tests/test_gpu_winograd44.py)


def test_edge_magnitudes(dev):
    """Operands at the ends of fp32's range (csrc/split_bf16.h, "Edge magnitudes"): zeros stay exact; 1e-30 keeps full accuracy; 1e-38 is carried with the 8-16 bits the
    bf16 denormals leave (the result is finite and within 1 % - the fp32 kernel is exact there); 3e37 is exact; a value that rounds to bf16's infinity, Inf and NaN give NaN
    in every output they touch (the fp32-MFMA kernel gives Inf / NaN there)."""
    from ssv_amd import _lib, ops
    t, c, k = 256, 256, 128
    g = torch.Generator(device=dev).manual_seed(5)
    base = torch.randn(1, t, c, device=dev, generator=g)
    w = torch.randn(1, k, c, device=dev, generator=g) * (1.0 / c) ** 0.5

    def run(a, arith="bf16x3", wt=w):
        y = torch.empty(1, t, k, device=dev)
        with ops.arithmetic(arith):
            ops._gemm_batched(1, t, c, k, a.contiguous(), wt.contiguous(), y)
        torch.cuda.synchronize()
        return y

    # +-0: exact zeros, no NaN from the residual arithmetic
    z = base.clone()
    z[0, :8] = 0.0
    z[0, 8:16] = -0.0
    y = run(z)
    assert (y[0, :16] == 0).all() and torch.isfinite(y).all()
    # 1e-30 (above 2^-109): full accuracy; 1e-38: degraded, finite, within 1 %
    for scale, bar in ((1e-30, 2e-6), (1e-38, 1e-2)):
        a = base * scale
        ref = torch.bmm(a.double(), w.double().transpose(1, 2))
        e = _rel(run(a), ref)
        assert e < bar, (scale, e)
        assert _rel(run(a, "f32"), ref) < 2e-6
    # 3e37 with small weights: exact pieces, finite sums
    a = torch.sign(base) * 3e37
    ws = w * 1e-3
    ref = torch.bmm(a.double(), ws.double().transpose(1, 2))
    assert _rel(run(a, wt=ws), ref) < 2e-6
    # a finite fp32 beyond bf16's largest value, Inf, NaN: NaN in the rows they touch, every other row untouched
    for bad in (3.4e38, float("inf"), float("nan")):
        a = base.clone()
        a[0, 3, 7] = bad
        y = run(a, wt=ws)
        assert torch.isnan(y[0, 3]).all(), bad
        assert torch.isfinite(y[0, :3]).all() and torch.isfinite(y[0, 4:]).all()
    yf = run(torch.where(torch.arange(c, device=dev) == 7, torch.tensor(float("inf"), device=dev), base[0, 3]).expand(1, t, c), "f32", wt=ws)
    assert not torch.isfinite(yf).any()            # the fp32-MFMA kernel: Inf (or NaN) there - non-finite either way


def test_planes_reconstruct_the_operand_exactly(dev):
    """ssv_split_planes: p0 + p1 + p2 == x bit for bit over normal magnitudes, and the planes are the round-to-nearest-even pieces the staging code forms."""
    from ssv_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.randn(1 << 16, device=dev, generator=g) * torch.exp(torch.randn(1 << 16, device=dev, generator=g) * 8)).contiguous()
    with ops.arithmetic("bf16x3"):
        pl = ops._planes(x)
    torch.cuda.synchronize()
    pieces = (pl.to(torch.int32) << 16).view(torch.float32).double()            # bf16 bits -> fp32 -> fp64
    assert torch.equal(pieces.sum(0), x.double())
    assert torch.equal(pl[0].view(torch.bfloat16).float(), x.to(torch.bfloat16).float())
