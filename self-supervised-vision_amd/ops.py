"""Tensor-level wrappers over the C ABI.  Activations are NHWC fp32 device tensors, filters are
[O,I,H,W] tensors in channels_last memory format (= OHWI in memory).  Every function only enqueues
kernels on the current HIP stream; torch provides allocation, nothing else."""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import ConvDesc, call, ptr, stream, workspace

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# HOW the GEMM-shaped launches multiply (include/ssv_hip.h: ssv_conv_desc.arithmetic; csrc/split_bf16.h).  "bf16x3" (default since round 6): every fp32 operand as three
# bf16 pieces, six exact piece products per fp32 product, fp32 accumulation on the bf16 matrix pipe - fp32 in, fp32 out, error against fp64 at or below the fp32-MFMA
# kernels' on every layer shape (tests/test_gpu_split.py).  "f32": v_mfma_f32_32x32x2_f32 on the fp32 operands, the arithmetic of rounds 1-5.
# SSV_ARITHMETIC=f32|bf16x3; bench.py reports both in one line.  Launches without a bf16-piece kernel (image stem, strided data gradient, grouped banks) run on
# fp32 MFMA either way.
ARITHMETIC = os.environ.get("SSV_ARITHMETIC", "bf16x3")
if ARITHMETIC not in ("bf16x3", "f32"):
    raise _lib.SsvError(f"SSV_ARITHMETIC must be bf16x3 or f32 (got {ARITHMETIC!r})")


class arithmetic:
    """``with ops.arithmetic("f32"): ...`` - the launches inside run on the named arithmetic (tests, bench.py's fp32-instruction leg)."""

    def __init__(self, name):
        if name not in ("bf16x3", "f32"):
            raise _lib.SsvError(f"arithmetic must be bf16x3 or f32 (got {name!r})")
        self.name = name

    def __enter__(self):
        global ARITHMETIC
        self.prev, ARITHMETIC = ARITHMETIC, self.name
        invalidate_weight_caches()
        return self

    def __exit__(self, *exc):
        global ARITHMETIC
        ARITHMETIC = self.prev
        invalidate_weight_caches()
        return False


def conv_desc(x_shape, w_shape, stride, pad, w=None):
    """``w``: the tensor the launch will be given as its weight operand (the filter, its transpose, a transformed filter) - in the bf16x3 arithmetic its pre-split
    planes ride in the descriptor (made once per weight, stream and step: `_planes`).  Launches whose operands are both activations (weight gradients) pass none."""
    n, h, w_, c = x_shape
    k, ci, r, s = w_shape
    if ci != c:
        raise _lib.SsvError(f"conv: input has {c} channels, filter expects {ci}")
    ho, wo = (h + 2 * pad - r) // stride + 1, (w_ + 2 * pad - s) // stride + 1
    d = ConvDesc(n, h, w_, c, k, r, s, stride, pad, ho, wo)
    if ARITHMETIC == "bf16x3":
        d.arithmetic = _lib.ARITH_BF16X3
        if w is not None and w.numel() % 8 == 0 and (c % 32 == 0 or k % 32 == 0):      # forward products contract over C, the strided data gradient over K
            d.w_planes = ptr(_planes(w))
    return d


def _ohwi(w):
    """The filter must be OHWI in memory: a channels_last [O,I,H,W] tensor (or any [O,I,1,1]/2-D matrix)."""
    if w.dim() == 2:
        if not w.is_contiguous():
            raise _lib.SsvError("linear weight must be contiguous")
        return w, (w.shape[0], w.shape[1], 1, 1)
    if not w.is_contiguous(memory_format=torch.channels_last):
        raise _lib.SsvError("conv filter must be in channels_last (OHWI) memory format")
    return w, tuple(w.shape)


# The conv kernels address every tensor through a buffer descriptor with a non-negative 32-bit byte offset, so one launch handles
# tensors below 2 GiB (check_desc in csrc/conv_mfma.hip).  Larger batches are split along N HERE: NHWC tensors are contiguous per
# sample, a chunk is a plain sub-range of every operand, and each chunk is its own launch on the same stream.
_MAX_ELEMS = (1 << 29) - (1 << 22)


def _batch_chunks(n, per_sample_elems, rows_per_sample=None):
    """[(n0, n1)] such that every chunk keeps all tensors below the per-launch limit.  ``rows_per_sample`` (output rows per sample):
    chunk rows are kept a multiple of 64 so that the 64-row statistics partials of the chunks concatenate to exactly the partials of
    the unsplit launch."""
    worst = max(per_sample_elems)
    if n * worst < _MAX_ELEMS:
        return [(0, n)]
    nc = (_MAX_ELEMS - 1) // worst
    if nc < 1:
        raise _lib.SsvError("a single sample exceeds the 2 GiB per-launch tensor limit of the convolution kernels")
    if rows_per_sample is not None:
        q = 64 // math.gcd(rows_per_sample, 64)
        if nc // q * q < 1:
            raise _lib.SsvError(f"the statistics epilogue needs chunks of a multiple of {q} samples ({rows_per_sample} output rows per sample) but only "
                                f"{nc} sample(s) fit below the 2 GiB per-launch tensor limit: run this layer without the fused statistics "
                                f"(SSV_NO_BN_STATS_FUSION=1) or with smaller images")
        nc = nc // q * q
    return [(s0, min(s0 + nc, n)) for s0 in range(0, n, nc)]


def _sub(t, n0, n1):
    return None if t is None else t[n0:n1]


def conv2d_fwd(x, w, stride=1, pad=0, bias=None, addend=None, groups=1):
    """``groups`` > 1: w is the DENSE block-diagonal bank of a grouped convolution (group_expand); every column tile then contracts over the
    channels of its own groups only."""
    _lib._dev(x, w, bias, addend)
    w, wshape = _ohwi(w)
    d = conv_desc(x.shape, wshape, stride, pad)
    y = _empty((d.N, d.Ho, d.Wo, d.K), x)
    for n0, n1 in _batch_chunks(d.N, (d.H * d.W * d.C, d.Ho * d.Wo * d.K)):
        dc = conv_desc((n1 - n0,) + tuple(x.shape[1:]), wshape, stride, pad, w=w)
        if groups > 1:
            call("ssv_conv2d_fwd_grouped", C.byref(dc), int(groups), ptr(x[n0:n1]), ptr(w), ptr(bias), ptr(_sub(addend, n0, n1)), ptr(y[n0:n1]), stream())
        else:
            call("ssv_conv2d_fwd", C.byref(dc), ptr(x[n0:n1]), ptr(w), ptr(bias), ptr(_sub(addend, n0, n1)), ptr(y[n0:n1]), stream())
    return y


def conv2d_fwd_stats(x, w, stride=1, pad=0, keep_v=False):
    """y = conv(x, w) plus the BatchNorm statistics partials of y from the same epilogue: (y, pmean, pm2[, rows per group]) with one partial
    per 64 output rows unless the third element says otherwise.  Returns None when the shape is outside the fused kernel's preconditions
    (C % 32, K % 4)."""
    _lib._dev(x, w)
    _, wshape = _ohwi(w)
    if (wshape[1] % 32 and wshape[1] != 4) or wshape[0] % 4:          # C == 4: the padded image stem
        return None
    d = conv_desc(x.shape, wshape, stride, pad)
    try:                     # huge samples with an odd row count: no 64-row-aligned chunk fits one launch - the caller runs the stand-alone statistics pass
        _batch_chunks(d.N, (d.H * d.W * d.C, d.Ho * d.Wo * d.K), rows_per_sample=d.Ho * d.Wo)
    except _lib.SsvError:
        return None
    y, part = conv2d_fwd_fused(x, w, stride, pad, in_affine=None, want_stats=True, keep_v=keep_v)
    return (y,) + tuple(part)


def conv2d_fwd_fused(x, w, stride=1, pad=0, in_affine=None, want_stats=True, keep_v=False):
    """y = conv(act(x), w) where act is the identity or, with ``in_affine = (scale, shift)``, relu(x * scale[c] + shift[c]) applied while
    the operand is staged (x is then the producer's RAW conv output: the activation itself is never written).  With ``want_stats``
    the epilogue also leaves the BatchNorm statistics partials of y: returns (y, (pmean, pm2[, rows per group]) | None).
    Stride-1 3x3 convolutions of the deep stages run through Winograd F(2x2, 3x3) (`use_winograd`); ``keep_v`` then leaves the transformed
    input on the result (``y._wino_v``) for `conv2d_wgrad`."""
    _lib._dev(x, w)
    w, wshape = _ohwi(w)
    d = conv_desc(x.shape, wshape, stride, pad)
    if use_winograd(wshape, stride, pad, x.shape, want_stats):
        y, part, v = wino_conv2d_fwd(x, w, in_affine=in_affine, want_stats=want_stats, keep_v=keep_v)
        if v is not None:
            y._wino_v = v                  # the transformed input: the weight gradient's operand (nn.conv hands it to conv2d_wgrad)
        return y, part
    if in_affine is None and not want_stats:
        return conv2d_fwd(x, w, stride, pad), None
    y = _empty((d.N, d.Ho, d.Wo, d.K), x)
    lib = _lib.load()
    groups = int(lib.ssv_conv2d_fwd_stats_groups(C.byref(d)))
    part = _empty((2, groups, d.K), x) if want_stats else None
    sc, sh = in_affine if in_affine is not None else (None, None)
    g0 = 0
    for n0, n1 in _batch_chunks(d.N, (d.H * d.W * d.C, d.Ho * d.Wo * d.K), rows_per_sample=d.Ho * d.Wo if want_stats else None):
        dc = conv_desc((n1 - n0,) + tuple(x.shape[1:]), wshape, stride, pad, w=w)
        gc = int(lib.ssv_conv2d_fwd_stats_groups(C.byref(dc)))
        call("ssv_conv2d_fwd_bnrelu_in_stats", C.byref(dc), ptr(x[n0:n1]), ptr(sc), ptr(sh), ptr(w), ptr(y[n0:n1]),
             ptr(part[0][g0:g0 + gc]) if want_stats else None, ptr(part[1][g0:g0 + gc]) if want_stats else None, stream())
        g0 += gc
    return y, (None if part is None else (part[0], part[1]))


def can_form_closing_sum(w_shape, stride, pad, groups=1):
    """Can this convolution form (and write) the closing activation of the previous residual unit while it stages its input?"""
    k, c, r, s_ = w_shape
    return groups == 1 and r == 1 and s_ == 1 and stride == 1 and pad == 0 and c % 32 == 0 and k % 4 == 0


def conv2d_fwd_sumin(x, res, scale, shift, res_affine, w, want_mask=True):
    """y = conv1x1(a, w) with a = relu(x * scale + shift + res) (``res_affine = (rscale, rshift)``: + res * rscale + rshift) formed while it is
    staged AND written out by the same kernel; the epilogue leaves the BatchNorm statistics partials of y.
    Returns (y, (pmean, pm2), a, mask | None) - a / mask bit-identical to `bn_apply(x, scale, shift, relu=True, residual=res, ...)`."""
    _lib._dev(x, res, w)
    w, wshape = _ohwi(w)
    d = conv_desc(x.shape, wshape, 1, 0)
    y = _empty((d.N, d.Ho, d.Wo, d.K), x)
    a = torch.empty_like(x)
    m, c = _rows(x)
    mask = torch.empty((m * c // 4,), dtype=torch.uint8, device=x.device) if want_mask else None
    lib = _lib.load()
    groups = int(lib.ssv_conv2d_fwd_stats_groups(C.byref(d)))
    part = _empty((2, groups, d.K), x)
    rs, rh = res_affine if res_affine is not None else (None, None)
    per = d.H * d.W * d.C
    g0 = 0
    for n0, n1 in _batch_chunks(d.N, (per, d.Ho * d.Wo * d.K), rows_per_sample=d.Ho * d.Wo):
        dc = conv_desc((n1 - n0,) + tuple(x.shape[1:]), wshape, 1, 0, w=w)
        gc = int(lib.ssv_conv2d_fwd_stats_groups(C.byref(dc)))
        call("ssv_conv2d_fwd_sumin_stats", C.byref(dc), ptr(x[n0:n1]), ptr(res[n0:n1]), ptr(scale), ptr(shift), ptr(rs), ptr(rh), ptr(w), ptr(y[n0:n1]),
             ptr(part[0][g0:g0 + gc]), ptr(part[1][g0:g0 + gc]), ptr(a[n0:n1]), None if mask is None else ptr(mask[n0 * per // 4:n1 * per // 4]), stream())
        g0 += gc
    return y, (part[0], part[1]), a, mask


def can_fuse_conv_input(cin, cout, groups=1):
    """Preconditions of the fused-input convolution kernels (forward and weight gradient)."""
    return groups == 1 and cin % 32 == 0 and cin <= 1024 and cout % 4 == 0


_WT_CACHE = {}          # (weight address, stream) -> (storage kept alive, transposed filter): one transpose per weight, stream and step
_WINO_U = {}            # (weight address, stream, shape, transposed?) -> (storage kept alive, Winograd-transformed filter), same lifetime
_PLANES = {}            # (operand address, stream, elements) -> (storage kept alive, its three bf16 planes), same lifetime


def invalidate_weight_caches():
    """Called by everything that mutates a GEMM operand in place (the optimizers' step() and zero_grad(), the EMA of target
    networks, MemoryBank pushes, checkpoint loads): a cached transposed filter must never outlive the weights it was made from."""
    _WT_CACHE.clear()
    _WINO_U.clear()
    _PLANES.clear()


def _planes(w):
    """The three bf16 planes of a weight operand (ssv_split_planes): [3][numel] bf16 as an int16 tensor, cached per HIP stream like the transposed filters."""
    n = w.numel()
    key = (w.data_ptr(), stream(), n, w._version)       # torch-level in-place edits of the operand move its version; the library's own updates call invalidate_weight_caches()
    hit = _PLANES.get(key)
    if hit is not None:
        return hit[1]
    if n % 8:
        raise _lib.SsvError(f"_planes: operand of {n} elements (needs a multiple of 8)")
    pl = torch.empty((3, n), dtype=torch.int16, device=w.device)
    call("ssv_split_planes", n, ptr(w), ptr(pl), stream())
    _PLANES[key] = (w.untyped_storage(), pl)
    return pl


def _transposed_filter(w, wshape):
    """wt[c][R-1-r][S-1-s][k] = w[k][r][s][c].  Cached per HIP stream (the transpose is ordered on that stream) until the next
    zero_grad(); the entry keeps the weight's storage alive, so an address can never come back as a different tensor."""
    key = (w.data_ptr(), stream(), wshape)
    hit = _WT_CACHE.get(key)
    if hit is not None:
        return hit[1]
    k, c, r, s_ = wshape
    wt = torch.empty((c, r, s_, k), dtype=torch.float32, device=w.device)
    call("ssv_filter_transpose", k, r, s_, c, ptr(w), ptr(wt), stream())
    _WT_CACHE[key] = (w.untyped_storage(), wt)
    return wt


class BnGateCtx:
    """What a convolution's data gradient needs to run the first half of the BatchNorm backward in its epilogue (include/ssv_hip.h,
    ssv_bn_gate): the BatchNorm's input x, its saved statistics, and the ReLU bit - byte mask or the forward's (scale, shift)."""
    __slots__ = ("x", "mean", "invstd", "mask", "scale", "shift", "x2", "mean2", "invstd2")

    def __init__(self, x, mean, invstd, mask=None, scale=None, shift=None, second=None):
        """``second = (x2, mean2, invstd2)``: the gated gradient is also the gradient w.r.t. the output of the projection shortcut's BatchNorm
        (input x2): the epilogue then reduces sum g * xhat2 as well (mask gates on stride-1 data gradients only; dropped otherwise)."""
        self.x, self.mean, self.invstd, self.mask, self.scale, self.shift = x, mean, invstd, mask, scale, shift
        self.x2, self.mean2, self.invstd2 = second if second is not None else (None, None, None)


class LazyGrad:
    """The gradient w.r.t. a convolution output x that sits behind a BatchNorm, NOT written to HBM: dx = A[c] * g + B[c] * (x - mean[c])
    + D[c] with g the (relu-gated) gradient w.r.t. the BatchNorm's output and coef = [A | mean | B | D] from `bn_bwd_coef`.  The producing
    convolution's weight and data gradient form it while they stage their dY operand (ssv_conv2d_wgrad_dyin / ssv_conv2d_fwd_dyin), so the
    element-wise pass of the BatchNorm backward does not run (1x1 / stride-1 convolutions: conv3 and the stride-1 projection shortcut of
    networks/resnet.py:66-75, the widest tensors of a residual unit)."""
    __slots__ = ("g", "x", "coef", "wino_vd")

    def __init__(self, g, x, coef):
        self.g, self.x, self.coef = g, x, coef
        self.wino_vd = None              # Winograd F(4x4) layers: the data gradient's transformed input, left by the weight gradient's pass over (g, x)

    @property
    def shape(self):
        return self.g.shape


class StridedGrad:
    """The data gradient of a 1x1 / stride-2 convolution (the projection shortcut of a stage entry, networks/resnet.py:131-135), COMPACT: ``t``
    [N, ceil(H/2), ceil(W/2), C] holds the gradient of the input pixels with even (h, w); every other pixel's is zero and is not stored.  The
    unit input's other contribution (conv1's data gradient) takes it as its addend (ssv_conv2d_fwd_*_s2add); `materialize` is the fallback."""
    __slots__ = ("t", "full_shape")

    def __init__(self, t, full_shape):
        self.t, self.full_shape = t, tuple(full_shape)

    @property
    def shape(self):
        return self.full_shape

    def materialize(self):
        full = fill_(torch.empty(self.full_shape, dtype=torch.float32, device=self.t.device), 0.0)
        full[:, ::2, ::2, :].copy_(self.t)            # rare path (a consumer that cannot take the compact form): torch's strided copy
        return full


def compact_s2_dgrad(dy, w, x_shape):
    """dx of a 1x1 / stride-2 / unpadded convolution on the subsampled grid: one dense GEMM, as a StridedGrad."""
    n, h, w_, c = x_shape
    ho, wo = dy.shape[1], dy.shape[2]
    if (ho, wo) != ((h + 1) // 2, (w_ + 1) // 2):
        raise _lib.SsvError("compact_s2_dgrad: output grid does not match a 1x1 / stride-2 convolution")
    return StridedGrad(conv2d_dgrad(dy, w, (n, ho, wo, c), 1, 0), x_shape)


def can_lazy_dy(w_shape, stride, pad):
    """Can the weight and data gradient of this convolution take their dY operand as a LazyGrad?"""
    k, c, r, s_ = w_shape
    return r == 1 and s_ == 1 and stride == 1 and pad == 0 and k % 32 == 0 and c % 4 == 0


def _dyin_struct(lg, n0, n1):
    return _lib.BnDyin(ptr(lg.x[n0:n1]), ptr(lg.coef))


def _gate_struct(gate, groups, channels, like, second=False):
    part = _empty((3 if second else 2, groups, channels), like)
    st = _lib.BnGate(ptr(gate.x), ptr(gate.scale), ptr(gate.shift), ptr(gate.mask), ptr(gate.mean), ptr(gate.invstd), ptr(part[0]), ptr(part[1]),
                     ptr(gate.x2) if second else None, ptr(gate.mean2) if second else None, ptr(gate.invstd2) if second else None, ptr(part[2]) if second else None)
    return st, part


def _gate_sub(gate, n0, n1):
    return BnGateCtx(gate.x[n0:n1], gate.mean, gate.invstd, mask=None if gate.mask is None else gate.mask[n0 * gate.x[0].numel() // 4:n1 * gate.x[0].numel() // 4],
                     scale=gate.scale, shift=gate.shift, second=None if gate.x2 is None else (gate.x2[n0:n1], gate.mean2, gate.invstd2))


def conv2d_dgrad(dy, w, x_shape, stride=1, pad=0, addend=None, out=None, gate=None, groups=1):
    """dx = conv_transpose(dy, w) (+ addend).  Stride-1 layers (every Linear, every 1x1 and 3x3 stride-1 convolution) are computed
    as the FORWARD convolution of dy with the transposed, 180-degree rotated filter: both GEMM operands are then k-contiguous
    rows for ds_read_b128, which the dgrad kernel (weights read in place, k-major) cannot have - measured 5-15 % faster.
    ``gate`` (BnGateCtx): dx is the gradient w.r.t. a BatchNorm + ReLU output and this call is its LAST contribution - the epilogue
    stores the relu-gated gradient and the partial sums of the BatchNorm backward; they come back as ``dx._gate_partials``
    (psum_g, psum_gx, groups).  Silently ungated when the shape is outside the gated kernels' preconditions.
    ``groups`` > 1: w is the dense block-diagonal bank of a grouped convolution (plain data gradient only: no gate, no formed-on-load operand)."""
    if groups > 1 and (gate is not None or isinstance(dy, LazyGrad) or isinstance(addend, StridedGrad)):
        raise _lib.SsvError("conv2d_dgrad: the grouped data gradient takes no gate / lazy operand")
    lazy = dy if isinstance(dy, LazyGrad) else None
    if lazy is not None:
        dy = lazy.g
    compact = addend if isinstance(addend, StridedGrad) else None
    if compact is not None:
        _, wsh = _ohwi(w)
        fits = (gate is not None and gate.mask is not None and stride == 1 and wsh[2] == 1 and wsh[3] == 1 and pad == 0 and wsh[1] >= 128
                and wsh[0] % 32 == 0 and wsh[1] % 4 == 0 and tuple(gate.x.shape) == tuple(x_shape) and dy[0].numel() * dy.shape[0] < _MAX_ELEMS
                and x_shape[0] * x_shape[1] * x_shape[2] * x_shape[3] < _MAX_ELEMS)
        if not fits:
            addend, compact = compact.materialize(), None
            out = addend
        else:
            addend, out = compact.t, None
    _lib._dev(dy, w, addend)
    w, wshape = _ohwi(w)
    k, c, r, s_ = wshape
    if lazy is not None and lazy.wino_vd is not None:            # a Winograd F(4x4) layer: the weight gradient's pass over (g, x) left this product's transformed input
        if not (addend is None and out is None and (gate is None or gate.x2 is None) and groups == 1):
            raise _lib.SsvError("conv2d_dgrad: a LazyGrad of a Winograd layer reached a data gradient with an addend / second gate target (nn.conv marks such layers only "
                                "behind a fused input chain, where neither exists)")
        return wino44_conv2d_dgrad(lazy, w, gate=gate)
    if lazy is not None and not can_lazy_dy(wshape, stride, pad):
        raise _lib.SsvError("conv2d_dgrad: a LazyGrad reached a convolution that cannot form it (the producer must check ops.can_lazy_dy)")
    dx = out if out is not None else _empty(tuple(x_shape), dy)
    dx.__dict__.pop("_gate_partials", None)            # an accumulated-into buffer never keeps the partial sums of its old content
    dx.__dict__.pop("_gate_partials_res", None)
    if gate is not None and (k % 32 or c % 4 or tuple(gate.x.shape) != tuple(dx.shape)):
        gate = None
    lib = _lib.load()
    n = dy.shape[0]
    if (lazy is None and addend is None and out is None and (gate is None or gate.x2 is None) and groups == 1
            and use_winograd((c, k, r, s_), stride, pad, dy.shape, False)):
        return wino_conv2d_dgrad(dy, w, gate=gate)
    chunks = _batch_chunks(n, (dy[0].numel(), dx[0].numel()))
    as_fwd = stride == 1 and r == s_ and r - 1 - pad >= 0 and k % 16 == 0 and c % 4 == 0
    if gate is not None and not as_fwd and stride > 8:
        gate = None
    wt = _transposed_filter(w, wshape) if as_fwd else None
    second = gate is not None and as_fwd and gate.x2 is not None and gate.mask is not None and tuple(gate.x2.shape) == tuple(dx.shape)
    parts = []
    for n0, n1 in chunks:
        dyc, dxc, adc = dy[n0:n1], dx[n0:n1], _sub(addend, n0, n1)
        if as_fwd:
            d = conv_desc(dyc.shape, (c, k, r, s_), 1, r - 1 - pad, w=wt)
            if (d.Ho, d.Wo) != (x_shape[1], x_shape[2]):
                raise _lib.SsvError("conv2d_dgrad: input shape does not match the stride-1 geometry")
            if compact is not None:         # one chunk (checked above); the compact stride-2 addend rides on the byte-mask gate epilogues
                h2, w2 = compact.t.shape[1], compact.t.shape[2]
                groups = int(lib.ssv_conv2d_fwd_gate_groups(C.byref(d)))
                st, part = _gate_struct(gate, groups, c, dy, second)
                parts.append(part)
                if lazy is not None:
                    dyin = _dyin_struct(lazy, n0, n1)
                    call("ssv_conv2d_fwd_dyin_s2add", C.byref(d), ptr(dyc), C.byref(dyin), ptr(wt), ptr(compact.t), h2, w2, ptr(dxc), C.byref(st), stream())
                else:
                    call("ssv_conv2d_fwd_gated_s2add", C.byref(d), ptr(dyc), ptr(wt), ptr(compact.t), h2, w2, ptr(dxc), C.byref(st), stream())
            elif lazy is not None:
                st = None
                if gate is not None:
                    groups = int(lib.ssv_conv2d_fwd_gate_groups(C.byref(d)))
                    st, part = _gate_struct(_gate_sub(gate, n0, n1), groups, c, dy, second)
                    parts.append(part)
                dyin = _dyin_struct(lazy, n0, n1)
                call("ssv_conv2d_fwd_dyin", C.byref(d), ptr(dyc), C.byref(dyin), ptr(wt), ptr(adc), ptr(dxc), None if st is None else C.byref(st), stream())
            elif gate is not None:
                groups = int(lib.ssv_conv2d_fwd_gate_groups(C.byref(d)))
                st, part = _gate_struct(_gate_sub(gate, n0, n1), groups, c, dy, second)
                call("ssv_conv2d_fwd_gated", C.byref(d), ptr(dyc), ptr(wt), ptr(adc), ptr(dxc), C.byref(st), stream())
                parts.append(part)
            elif groups > 1:          # the transposed bank is block-diagonal too (input and output channels of a group change places)
                call("ssv_conv2d_fwd_grouped", C.byref(d), int(groups), ptr(dyc), ptr(wt), None, ptr(adc), ptr(dxc), stream())
            else:
                call("ssv_conv2d_fwd", C.byref(d), ptr(dyc), ptr(wt), None, ptr(adc), ptr(dxc), stream())
        else:
            d = conv_desc(dxc.shape, wshape, stride, pad, w=w if groups == 1 else None)
            if gate is not None:
                groups = int(lib.ssv_conv2d_dgrad_gate_groups(C.byref(d)))
                st, part = _gate_struct(_gate_sub(gate, n0, n1), groups, c, dy)
                call("ssv_conv2d_dgrad_gated", C.byref(d), ptr(dyc), ptr(w), ptr(adc), ptr(dxc), C.byref(st), stream())
                parts.append(part)
            elif groups > 1:
                call("ssv_conv2d_dgrad_grouped", C.byref(d), int(groups), ptr(dyc), ptr(w), ptr(adc), ptr(dxc), stream())
            else:
                call("ssv_conv2d_dgrad", C.byref(d), ptr(dyc), ptr(w), ptr(adc), ptr(dxc), stream())
    if parts:
        part = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)      # partial SUMS: any grouping of the rows adds up the same
        dx._gate_partials = (part[0], part[1], part.shape[1])
        if part.shape[0] == 3:                 # the same g against the projection shortcut's BatchNorm input
            dx._gate_partials_res = (part[0], part[2], part.shape[1])
    return dx


FUSE_BIAS_GRAD = os.environ.get("SSV_NO_BIAS_GRAD_FUSION", "0") != "1"      # diagnostic switch: bias gradients by the stand-alone column-sum pass


def conv2d_wgrad(x, dy, w_like, dw, stride=1, pad=0, accumulate=True, in_affine=None, wino_v=None, groups=1, dbias=None, dgrad_follows=None):
    """dw (+)= wgrad.  ``dw`` has the memory layout of ``w_like`` (OHWI).  ``in_affine = (scale, shift)``: x is a raw conv output and
    the operand is relu(x * scale + shift), formed on load (the fused chain's never-materialised activation).  ``wino_v``: the transformed
    input the Winograd forward of this convolution kept - the weight gradient is then 16 batched GEMMs on it (x / in_affine are not read).
    ``groups`` > 1: ``dw`` is the dense block-diagonal layout of a grouped convolution's bank and ONLY its diagonal blocks are defined afterwards
    (what group_extract reads); tiles no group touches are skipped.
    ``dgrad_follows`` (Winograd F(4x4) layers): will a PLAIN data gradient of the same dy follow (no addend, no accumulation target)?  Then the dY transform also
    writes that product's transformed input (one pass over dy); False spares the extra 2.25x write when the data gradient will take another kernel; None = decide by
    the dispatch rule alone.
    ``dbias``: the layer's bias gradient, (+)= the column sums of dy - taken from the same pass over dy when the layer is a Linear / 1x1 / stride-1
    one (the weight-gradient workgroups of column tile 0 sum the rows they stage), by the stand-alone column-sum kernel otherwise."""
    if dbias is not None:
        _, wsh = _ohwi(w_like)
        fused = (FUSE_BIAS_GRAD and groups == 1 and wino_v is None and in_affine is None and not isinstance(dy, LazyGrad)
                 and wsh[2] == 1 and wsh[3] == 1 and stride == 1 and pad == 0 and wsh[0] % 4 == 0 and wsh[1] % 4 == 0)
        if isinstance(dy, LazyGrad):
            raise _lib.SsvError("conv2d_wgrad: a bias gradient needs the materialised output gradient (a biased convolution is never followed by a fused BatchNorm)")
        if not fused:
            colsum(dy, dbias, accumulate=accumulate)
            return conv2d_wgrad(x, dy, w_like, dw, stride, pad, accumulate, in_affine, wino_v, groups)
        _lib._dev(x, dy, dw, dbias)
        lib = _lib.load()
        for i, (n0, n1) in enumerate(_batch_chunks(x.shape[0], (x[0].numel(), dy[0].numel()))):
            d = conv_desc(x[n0:n1].shape, wsh, stride, pad)
            ws = workspace.get(lib.ssv_conv2d_wgrad_bias_workspace_bytes(C.byref(d)), x.device)
            call("ssv_conv2d_wgrad_bias", C.byref(d), ptr(x[n0:n1]), ptr(dy[n0:n1]), ptr(dw), ptr(dbias), int(accumulate or i > 0), ptr(ws), ws.numel(), stream())
        return dw
    if groups > 1:
        if isinstance(dy, LazyGrad) or in_affine is not None or wino_v is not None:
            raise _lib.SsvError("conv2d_wgrad: the grouped weight gradient takes plain operands")
        _lib._dev(x, dy, dw)
        _, wshape = _ohwi(w_like)
        lib = _lib.load()
        for i, (n0, n1) in enumerate(_batch_chunks(x.shape[0], (x[0].numel(), dy[0].numel()))):
            d = conv_desc(x[n0:n1].shape, wshape, stride, pad)
            ws = workspace.get(lib.ssv_conv2d_wgrad_grouped_workspace_bytes(C.byref(d), int(groups)), x.device)
            call("ssv_conv2d_wgrad_grouped", C.byref(d), int(groups), ptr(x[n0:n1]), ptr(dy[n0:n1]), ptr(dw), int(accumulate or i > 0), ptr(ws), ws.numel(), stream())
        return dw
    if wino_v is not None and (not isinstance(dy, LazyGrad) or (wino_v.shape[0] == 36 and WINOGRAD44_DY_BOTH)):
        return wino_conv2d_wgrad(wino_v, dy, w_like, dw, accumulate=accumulate, dgrad_follows=dgrad_follows)
    lazy = dy if isinstance(dy, LazyGrad) else None
    if lazy is not None:
        dy = lazy.g
    _lib._dev(x, dy, dw)
    _, wshape = _ohwi(w_like)
    if lazy is not None and not can_lazy_dy(wshape, stride, pad):
        raise _lib.SsvError("conv2d_wgrad: a LazyGrad reached a convolution that cannot form it (the producer must check ops.can_lazy_dy)")
    lib = _lib.load()
    sc, sh = in_affine if in_affine is not None else (None, None)
    for i, (n0, n1) in enumerate(_batch_chunks(x.shape[0], (x[0].numel(), dy[0].numel()))):
        xc, dyc = x[n0:n1], dy[n0:n1]
        d = conv_desc(xc.shape, wshape, stride, pad)
        ws = workspace.get(lib.ssv_conv2d_wgrad_workspace_bytes(C.byref(d)), x.device)
        if lazy is not None:
            dyin = _dyin_struct(lazy, n0, n1)
            call("ssv_conv2d_wgrad_dyin", C.byref(d), ptr(xc), ptr(sc), ptr(sh), ptr(dyc), C.byref(dyin), ptr(dw), int(accumulate or i > 0), ptr(ws), ws.numel(), stream())
            continue
        call("ssv_conv2d_wgrad_bnrelu_in", C.byref(d), ptr(xc), ptr(sc), ptr(sh), ptr(dyc), ptr(dw), int(accumulate or i > 0), ptr(ws), ws.numel(), stream())
    return dw


def _rows(x):
    c = x.shape[-1]
    return x.numel() // c, c


def _rows_per_group(partials):
    """Statistics partials are (pmean, pm2) over groups of 64 output rows (the GEMM epilogues) or (pmean, pm2, rows per group) (the Winograd
    output transform on odd maps: one image per group)."""
    return int(partials[2]) if len(partials) > 2 else 64


def bn_train_fwd(x, gamma, beta, running_mean, running_var, nbt, relu=False, residual=None,
                 eps=BN_EPS, momentum=BN_MOMENTUM, want_mask=False, skip_mask=False, partials=None):
    """Returns (y, mean, invstd[, relu_mask]).  relu_mask: uint8, one byte per four channels, for the backward.
    ``partials`` = (pmean, pm2) from conv2d_fwd_stats: the statistics pass over x is skipped."""
    _lib._dev(x, gamma, beta, residual)
    m, c = _rows(x)
    y = torch.empty_like(x)
    mean, invstd = _empty((c,), x), _empty((c,), x)
    mask = torch.empty((m * c // 4,), dtype=torch.uint8, device=x.device) if (want_mask and relu and not skip_mask) else None
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), x.device)
    if partials is not None:
        call("ssv_bn_train_fwd_partials", m, c, ptr(x), ptr(partials[0]), ptr(partials[1]), _rows_per_group(partials), ptr(gamma), ptr(beta), ptr(residual), int(relu), eps, momentum,
             ptr(running_mean), ptr(running_var), ptr(nbt), ptr(y), ptr(mask), ptr(mean), ptr(invstd), ptr(ws), ws.numel(), stream())
        return (y, mean, invstd, mask) if want_mask else (y, mean, invstd)
    call("ssv_bn_train_fwd", m, c, ptr(x), ptr(gamma), ptr(beta), ptr(residual), int(relu), eps, momentum,
         ptr(running_mean), ptr(running_var), ptr(nbt), ptr(y), ptr(mask), ptr(mean), ptr(invstd), ptr(ws), ws.numel(), stream())
    return (y, mean, invstd, mask) if want_mask else (y, mean, invstd)


def bn_stats_finalize(x_shape_rows, c, partials, gamma, beta, running_mean, running_var, nbt, eps=BN_EPS, momentum=BN_MOMENTUM):
    """Statistics partials of a conv output -> (mean, invstd, scale, shift) + running statistics: BatchNorm without its apply pass."""
    m = int(x_shape_rows)
    dev = gamma.device
    stats = torch.empty((4, c), dtype=torch.float32, device=dev)            # mean | invstd | scale | shift
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), dev)
    call("ssv_bn_stats_finalize", m, c, ptr(partials[0]), ptr(partials[1]), _rows_per_group(partials), ptr(gamma), ptr(beta), eps, momentum,
         ptr(running_mean), ptr(running_var), ptr(nbt), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), ptr(ws), ws.numel(), stream())
    return stats[0], stats[1], stats[2], stats[3]


def bn_apply(x, scale, shift, relu=False, residual=None, res_affine=None, want_mask=False):
    """y = relu?(x * scale + shift (+ residual | + residual * res_scale + res_shift)); returns (y, mask | None)."""
    _lib._dev(x, scale, shift, residual)
    m, c = _rows(x)
    y = torch.empty_like(x)
    mask = torch.empty((m * c // 4,), dtype=torch.uint8, device=x.device) if (want_mask and relu) else None
    rs, rh = res_affine if res_affine is not None else (None, None)
    call("ssv_bn_apply", m, c, ptr(x), ptr(scale), ptr(shift), ptr(residual), ptr(rs), ptr(rh), int(relu), ptr(y), ptr(mask), stream())
    return y, mask


def bn_relu_bwd_affine(dy, x, gamma, mean, invstd, scale, shift, dgamma, dbeta, accumulate=True):
    """Backward of a BatchNorm + ReLU whose output was never materialised: the gate is x * scale + shift > 0."""
    _lib._dev(dy, x)
    m, c = _rows(x)
    dx = torch.empty_like(x)
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), x.device)
    call("ssv_bn_relu_bwd_affine", m, c, ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(dx),
         ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), ws.numel(), stream())
    return dx


def bn_bwd_from_partials(g, x, gamma, mean, invstd, partials, dgamma, dbeta, accumulate=True):
    """Second half of the BatchNorm backward behind a gated convolution (g already relu-gated, partial sums given)."""
    _lib._dev(g, x)
    m, c = _rows(x)
    dx = torch.empty_like(x)
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), x.device)
    call("ssv_bn_bwd_from_partials", m, c, ptr(g), ptr(x), ptr(gamma), ptr(mean), ptr(invstd), ptr(partials[0]), ptr(partials[1]), int(partials[2]),
         ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), ws.numel(), stream())
    return dx


def bn_bwd_coef(x, gamma, mean, invstd, partials, dgamma, dbeta, accumulate=True):
    """The merge of `bn_bwd_from_partials` without its element-wise pass: dgamma / dbeta and the [4, C] coefficients of a LazyGrad."""
    _lib._dev(x, gamma)
    m, c = _rows(x)
    coef = torch.empty((4, c), dtype=torch.float32, device=x.device)
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), x.device)
    call("ssv_bn_bwd_coef", m, c, ptr(gamma), ptr(mean), ptr(invstd), ptr(partials[0]), ptr(partials[1]), int(partials[2]),
         ptr(coef), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), ws.numel(), stream())
    return coef


def bn_train_bwd(dy, y, x, gamma, mean, invstd, relu, dgamma, dbeta, want_dres=False, accumulate=True, relu_mask=None):
    _lib._dev(dy, y, x)
    m, c = _rows(x)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), x.device)
    call("ssv_bn_train_bwd", m, c, ptr(dy), ptr(y), ptr(relu_mask), ptr(x), ptr(gamma), ptr(mean), ptr(invstd), int(relu),
         ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), ws.numel(), stream())
    return dx, dres


def colsum(x, out, accumulate=True):
    m, c = _rows(x)
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), x.device)
    call("ssv_colsum", m, c, ptr(x), ptr(out), int(accumulate), ptr(ws), ws.numel(), stream())
    return out


def bn_relu_maxpool_fwd(y, scale, shift, keep_xmax=False):
    """maxpool3x3s2(relu(y * scale + shift)) of a raw conv output [N,H,W,C] in one pass: returns (pooled, argmax slots[, xmax]).
    ``keep_xmax``: also the raw conv output at every window's arg-max pixel - what lets the backward reduce at the pooled resolution."""
    _lib._dev(y, scale, shift)
    n, h, w, c = y.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = _empty((n, ho, wo, c), y)
    am = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=y.device)
    xmax = _empty((n, ho, wo, c), y) if keep_xmax else None
    call("ssv_bn_relu_maxpool_fwd", n, h, w, c, ptr(y), ptr(scale), ptr(shift), ptr(out), ptr(am), ptr(xmax), stream())
    return (out, am, xmax) if keep_xmax else (out, am)


def bn_relu_maxpool_bwd(dpool, am, y, gamma, mean, invstd, scale, shift, dgamma, dbeta, accumulate=True, xmax=None):
    """Gradient w.r.t. the raw conv output y through maxpool, ReLU and BatchNorm, plus dgamma / dbeta.  ``xmax`` (from the forward): the
    reduction pass runs over the pooled positions instead of the full-resolution map (equal to rounding)."""
    _lib._dev(dpool, am, y, xmax)
    n, h, w, c = y.shape
    dy = torch.empty_like(y)
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(n * h * w, c), y.device)
    call("ssv_bn_relu_maxpool_bwd", n, h, w, c, ptr(dpool), ptr(am), ptr(y), ptr(xmax), ptr(gamma), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
         ptr(dy), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), ws.numel(), stream())
    return dy


def maxpool_fwd(x):
    n, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = _empty((n, ho, wo, c), x)
    am = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    call("ssv_maxpool3x3s2_fwd", n, h, w, c, ptr(x), ptr(y), ptr(am), stream())
    return y, am


def maxpool_bwd(dy, am, x_shape):
    n, h, w, c = x_shape
    dx = _empty(tuple(x_shape), dy)
    call("ssv_maxpool3x3s2_bwd", n, h, w, c, ptr(dy), ptr(am), ptr(dx), stream())
    return dx


def gap_fwd(x):
    n, h, w, c = x.shape
    y = _empty((n, c), x)
    call("ssv_gap_fwd", n, h * w, c, ptr(x), ptr(y), stream())
    return y


def gap_bwd(dy, x_shape):
    n, h, w, c = x_shape
    dx = _empty(tuple(x_shape), dy)
    call("ssv_gap_bwd", n, h * w, c, ptr(dy), ptr(dx), stream())
    return dx


def nchw_to_nhwc(x):
    """[N,C,H,W] (any strides) -> contiguous NHWC.  A channels_last input already IS NHWC in memory."""
    _lib._dev(x)
    n, c, h, w = x.shape
    if x.is_contiguous(memory_format=torch.channels_last) and c > 1:
        return x.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        raise _lib.SsvError("image batch must be contiguous NCHW or channels_last")
    y = _empty((n, h, w, c), x)
    call("ssv_nchw_to_nhwc", n, c, h, w, ptr(x), ptr(y), stream())
    return y


def l2norm_fwd(z, normalize=True, ldo=None, eps=1e-12, out=None):
    _lib._dev(z)
    rows, d = z.shape
    ldo = d if ldo is None else ldo
    zhat = out if out is not None else _empty((rows, ldo), z)
    inv = _empty((rows,), z)
    call("ssv_l2norm_fwd", rows, d, ptr(z), int(normalize), eps, ptr(zhat), ldo, ptr(inv), stream())
    return zhat, inv


def l2norm_bwd(zhat, inv, dzhat, d, normalize=True):
    rows = zhat.shape[0]
    dz = _empty((rows, d), zhat)
    call("ssv_l2norm_bwd", rows, d, ptr(zhat), zhat.shape[1], ptr(inv), ptr(dzhat), dzhat.shape[1], int(normalize), ptr(dz), stream())
    return dz


def ema_(dst_flat, src_flat, momentum):
    """dst = momentum * dst + (1 - momentum) * src[:len(dst)]: the target arena mirrors a PREFIX of the online arena (same offsets;
    BYOL's online network has the predictor behind it)."""
    _lib._dev(dst_flat, src_flat)
    if dst_flat.numel() > src_flat.numel():
        raise _lib.SsvError("ema_: the source arena is shorter than the target")
    invalidate_weight_caches()
    call("ssv_ema", dst_flat.numel(), ptr(dst_flat), ptr(src_flat), float(momentum), stream())
    return dst_flat


def scale_(x, factor_dev):
    call("ssv_scale", x.numel(), ptr(x), ptr(factor_dev), stream())
    return x


def fill_(x, value):
    call("ssv_fill", x.numel(), ptr(x), float(value), stream())
    return x


def add_(dst, src):
    call("ssv_add", dst.numel(), ptr(dst), ptr(src), stream())
    return dst


def ntxent_splits(nglob, b):
    """Column splits of the NT-Xent row kernels at this shape: the library's choice (ssv_ntxent_default_splits), or SSV_NTXENT_SPLITS
    (diagnostic switch: 1 = the unsplit kernels)."""
    forced = os.environ.get("SSV_NTXENT_SPLITS")
    if forced:
        return max(1, min(int(forced), (2 * nglob + 31) // 32, 64))
    return int(_lib.load().ssv_ntxent_default_splits(nglob, b))


def _ntxent_ws(zall, b, splits):
    if splits <= 1:
        return None, 0
    nbytes = int(_lib.load().ssv_ntxent_split_workspace_bytes(b, zall.shape[1], splits))
    return workspace.get(nbytes, zall.device), nbytes


def ntxent_fwd(zall, nglob, b, seg0, inv_temp, splits=None):
    """Row log-sum-exp and positive logit of this rank's 2*b rows against the gathered [2*nglob, ld] matrix."""
    _lib._dev(zall)
    lse = _empty((2 * b,), zall)
    pos = _empty((2 * b,), zall)
    splits = ntxent_splits(nglob, b) if splits is None else splits
    ws, nbytes = _ntxent_ws(zall, b, splits)
    call("ssv_ntxent_fwd_split", nglob, b, seg0, zall.shape[1], ptr(zall), float(inv_temp), ptr(lse), ptr(pos), splits, ptr(ws), nbytes, stream())
    return lse, pos


def ntxent_gram_fwd(gram, nglob, b, seg0, inv_temp):
    """(lse, pos) of this rank's 2*b rows from the materialised Gram block [2*b, lds] (wide embeddings)."""
    _lib._dev(gram)
    lse, pos = _empty((2 * b,), gram), _empty((2 * b,), gram)
    call("ssv_ntxent_gram_fwd", nglob, b, seg0, gram.shape[1], ptr(gram), float(inv_temp), ptr(lse), ptr(pos), stream())
    return lse, pos


def ntxent_gram_weights(gram, lse_all, nglob, b, seg0, inv_temp, gscale):
    """In place: the Gram block becomes the weight matrix W' whose product with Z_all is dZ."""
    _lib._dev(gram, lse_all)
    call("ssv_ntxent_gram_weights", nglob, b, seg0, gram.shape[1], ptr(gram), ptr(lse_all), float(inv_temp), float(gscale), stream())
    return gram


def ntxent_loss(lse, pos, scale):
    loss = torch.empty((), dtype=torch.float32, device=lse.device)
    call("ssv_ntxent_loss", lse.numel(), ptr(lse), ptr(pos), float(scale), ptr(loss), stream())
    return loss


def ntxent_bwd(zall, lse_all, nglob, b, seg0, inv_temp, gscale, splits=None):
    _lib._dev(zall, lse_all)
    dz = _empty((2 * b, zall.shape[1]), zall)
    splits = ntxent_splits(nglob, b) if splits is None else splits
    ws, nbytes = _ntxent_ws(zall, b, splits)
    call("ssv_ntxent_bwd_split", nglob, b, seg0, zall.shape[1], ptr(zall), ptr(lse_all), float(inv_temp), float(gscale), ptr(dz),
         splits, ptr(ws), nbytes, stream())
    return dz


def mse_pair(o1, o2, t1, t2, scale):
    """loss = scale * (|o1 - t2|^2 + |o2 - t1|^2) summed over all elements; returns (loss 0-d, dloss/do1, dloss/do2)."""
    _lib._dev(o1, o2, t1, t2)
    n = o1.numel()
    do1, do2 = torch.empty_like(o1), torch.empty_like(o2)
    loss = torch.empty((), dtype=torch.float32, device=o1.device)
    ws = _lib.workspace.get(_lib.load().ssv_reduce_workspace_bytes(n), o1.device)
    call("ssv_mse_pair_fwd_bwd", n, ptr(o1), ptr(o2), ptr(t1), ptr(t2), float(scale), ptr(loss), ptr(do1), ptr(do2), ptr(ws), ws.numel(), stream())
    return loss, do1, do2


def barlow_cgrad(craw, inv_b, lmbda):
    """C = craw * inv_b; loss = sum_ii (C-1)^2 + lmbda * sum_{i!=j} C^2; returns (loss 0-d, G = dloss/dcraw [D,D])."""
    _lib._dev(craw)
    d = craw.shape[0]
    g = torch.empty_like(craw)
    loss = torch.empty((), dtype=torch.float32, device=craw.device)
    ws = _lib.workspace.get(_lib.load().ssv_reduce_workspace_bytes(d * d), craw.device)
    call("ssv_barlow_cgrad", d, ptr(craw), float(inv_b), float(lmbda), ptr(loss), ptr(g), ptr(ws), ws.numel(), stream())
    return loss, g


def knn_label_agreement(z, labels, k):
    """Number of (query, neighbour) pairs with equal labels among each row's k nearest neighbours by inner product
    (best hit dropped), as a python int.  z [n,d] fp32 device, labels [n] int32 device."""
    _lib._dev(z, labels)
    n, d = z.shape
    if d % 4:
        z = torch.nn.functional.pad(z, (0, 4 - d % 4))
        d = z.shape[1]
    z = z.contiguous()
    count = torch.empty((1,), dtype=torch.int64, device=z.device)
    arith = _lib.ARITH_BF16X3 if ARITHMETIC == "bf16x3" else _lib.ARITH_F32_MFMA
    ws = torch.empty(_lib.load().ssv_knn_workspace_bytes_arith(n, d, arith), dtype=torch.uint8, device=z.device)    # up to ~1 GB: not kept in the training scratch
    call("ssv_knn_label_agreement_arith", n, d, ptr(z), ptr(labels), int(k), ptr(count), arith, ptr(ws), ws.numel(), stream())
    return int(count.item())


# ------------------------------------------------------------------------------------------- ViT / DINO pieces
LN_EPS = 1e-5
ATTENTION_BF16X3 = True      # the attention forward's two products in the bf16x3 arithmetic when ops.ARITHMETIC says so (the backward stays on the fp32 instruction)


def layernorm_fwd(x, gamma, beta, addend=None, eps=LN_EPS):
    """y = LayerNorm(x) * gamma + beta (+ addend) over the last axis of a dense [M, C] matrix; returns (y, mean, invstd)."""
    _lib._dev(x, gamma, beta, addend)
    m, c = _rows(x)
    y = torch.empty_like(x)
    mean, invstd = _empty((m,), x), _empty((m,), x)
    call("ssv_layernorm_fwd", m, c, ptr(x), ptr(gamma), ptr(beta), ptr(addend), float(eps), ptr(y), ptr(mean), ptr(invstd), stream())
    return y, mean, invstd


def layernorm_bwd(dy, x, gamma, mean, invstd, dgamma, dbeta, dx_addend=None, accumulate=True):
    _lib._dev(dy, x, dx_addend)
    m, c = _rows(x)
    dx = dx_addend if dx_addend is not None else torch.empty_like(x)      # in-place accumulate into the addend when given
    ws = workspace.get(_lib.load().ssv_layernorm_workspace_bytes(m, c), x.device)
    call("ssv_layernorm_bwd", m, c, ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(invstd), ptr(dx_addend), ptr(dx), ptr(dgamma), ptr(dbeta),
         int(accumulate), ptr(ws), ws.numel(), stream())
    return dx


def gelu_fwd(x):
    _lib._dev(x)
    y = torch.empty_like(x)
    call("ssv_gelu_fwd", x.numel(), ptr(x), ptr(y), stream())
    return y


def gelu_bwd(x, dy):
    _lib._dev(x, dy)
    dx = torch.empty_like(x)
    call("ssv_gelu_bwd", x.numel(), ptr(x), ptr(dy), ptr(dx), stream())
    return dx


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.SsvError("attention operands must be [M, heads*dh] matrices with unit column stride")
    return t.stride(0)


def attention_fwd(q, k, v, batch, tokens, heads):
    """softmax(q k^T / sqrt(dh)) v per (image, head); q/k/v: [batch*tokens, heads*dh] (may be column slices of one matrix)."""
    _lib._dev(q, k, v)
    hid = q.shape[1]
    dh = hid // heads
    if not (_ld(q) == _ld(k) == _ld(v)):
        raise _lib.SsvError("attention: q, k, v must share one row stride")
    o = torch.empty((batch * tokens, hid), dtype=torch.float32, device=q.device)
    lse = torch.empty((batch, heads, tokens), dtype=torch.float32, device=q.device)
    call("ssv_attention_fwd_arith", batch, tokens, heads, dh, ptr(q), ptr(k), ptr(v), _ld(q), dh ** -0.5, ptr(o), hid, ptr(lse),
         _lib.ARITH_BF16X3 if (ARITHMETIC == "bf16x3" and ATTENTION_BF16X3) else _lib.ARITH_F32_MFMA, stream())
    return o, lse


def attention_bwd(q, k, v, o, dout, lse, batch, tokens, heads, out=None):
    """Returns (dq, dk, dv); with ``out`` = a [M, 3*hidden] matrix they are its three column blocks."""
    _lib._dev(q, k, v, o, dout, lse)
    hid = q.shape[1]
    dh = hid // heads
    if out is None:
        dq, dk, dv = (torch.empty((batch * tokens, hid), dtype=torch.float32, device=q.device) for _ in range(3))
    else:
        dq, dk, dv = out[:, :hid], out[:, hid:2 * hid], out[:, 2 * hid:]
    delta = torch.empty_like(lse)
    call("ssv_attention_bwd", batch, tokens, heads, dh, ptr(q), ptr(k), ptr(v), _ld(q), dh ** -0.5, ptr(o), ptr(dout), _ld(o), ptr(lse),
         ptr(delta), ptr(dq), ptr(dk), ptr(dv), _ld(dq), stream())
    return dq, dk, dv


def vit_embed_fwd(img_nhwc, cls, pos, patch):
    """[B,H,W,3] image -> token matrix [B*T, 3*patch^2 + E] (cls row first, positional embedding concatenated)."""
    _lib._dev(img_nhwc, cls, pos)
    b, h, w, _ = img_nhwc.shape
    e = pos.shape[1]
    t = (h // patch) * (w // patch) + 1
    if t > pos.shape[0]:
        raise _lib.SsvError(f"vit_embed: {t} tokens but only {pos.shape[0]} positional embeddings")
    tok = torch.empty((b * t, 3 * patch * patch + e), dtype=torch.float32, device=img_nhwc.device)
    call("ssv_vit_embed_fwd", b, h, w, patch, e, ptr(img_nhwc), ptr(cls), ptr(pos), ptr(tok), stream())
    return tok, t


def vit_embed_bwd(dtok, batch, tokens, p3, e, dcls, dpos, accumulate=True):
    _lib._dev(dtok, dcls, dpos)
    call("ssv_vit_embed_bwd", batch, tokens, p3, e, ptr(dtok), ptr(dcls), ptr(dpos), int(accumulate), stream())


def weightnorm_fwd(g, v):
    _lib._dev(g, v)
    w = torch.empty_like(v)
    inv = _empty((v.shape[0],), v)
    call("ssv_weightnorm_fwd", v.shape[0], v.shape[1], ptr(g), ptr(v), ptr(w), ptr(inv), stream())
    return w, inv


def weightnorm_bwd(dw, g, v, inv, dg, dv, accumulate=True):
    _lib._dev(dw, g, v, inv, dg, dv)
    call("ssv_weightnorm_bwd", v.shape[0], v.shape[1], ptr(dw), ptr(g), ptr(v), ptr(inv), ptr(dg), ptr(dv), int(accumulate), stream())


def dino_loss(teacher, student, center, temp_s, temp_t, weight, loss, accumulate):
    """teacher [bs,2,K], student [bs,V,K] (dense), center [K]; loss (0-d device tensor) (+)= weighted loss; returns dstudent."""
    _lib._dev(teacher, student, center, loss)
    bs, v, k = student.shape
    if tuple(teacher.shape) != (bs, 2, k) or center.numel() != k:
        raise _lib.SsvError(f"dino_loss: teacher {tuple(teacher.shape)} / center {tuple(center.shape)} do not match student {tuple(student.shape)}")
    d = torch.empty_like(student)
    ws = workspace.get(_lib.load().ssv_dino_loss_workspace_bytes(bs, v, k), student.device)
    call("ssv_dino_loss", bs, v, k, ptr(teacher), ptr(student), ptr(center), float(temp_s), float(temp_t), float(weight), ptr(loss),
         int(accumulate), ptr(d), ptr(ws), ws.numel(), stream())
    return d


def dino_center_update(center, t1, t2, momentum):
    _lib._dev(center, t1, t2)
    k = center.numel()
    call("ssv_dino_center_update", k, t1.numel() // k, ptr(t1), 0 if t2 is None else t2.numel() // k, ptr(t2), float(momentum), ptr(center), stream())
    return center


def multicrop_params(batch, hs, ws_, ncrop, view_base, scale, seed, step, sample_ids=None, sample0=0, device=None):
    boxes = torch.empty((batch, ncrop, 4), dtype=torch.int32, device=device if sample_ids is None else sample_ids.device)
    call("ssv_multicrop_params", batch, hs, ws_, ncrop, view_base, float(scale[0]), float(scale[1]), int(seed), int(step), ptr(sample_ids),
         int(sample0), ptr(boxes), stream())
    return boxes


def multicrop(views_nhwc, boxes, size):
    """views [B,Hs,Ws,3] float, boxes [B,ncrop,4] int32 -> [B,ncrop,Ho,Wo,3] (crop + bicubic resize)."""
    _lib._dev(views_nhwc, boxes)
    b, hs, ws_, _ = views_nhwc.shape
    ncrop = boxes.shape[1]
    out = torch.empty((b, ncrop, size[0], size[1], 3), dtype=torch.float32, device=views_nhwc.device)
    call("ssv_multicrop", b, hs, ws_, ptr(views_nhwc), ncrop, ptr(boxes), size[0], size[1], ptr(out), stream())
    return out


# ------------------------------------------------------------------------------------------- sibling algorithms' losses
def negdot_pair(o1, o2, t1, t2, scale):
    """loss = -scale * (sum o1*t2 + sum o2*t1); returns (loss 0-d, dloss/do1, dloss/do2)."""
    _lib._dev(o1, o2, t1, t2)
    n = o1.numel()
    do1, do2 = torch.empty_like(o1), torch.empty_like(o2)
    loss = torch.empty((), dtype=torch.float32, device=o1.device)
    ws = workspace.get(_lib.load().ssv_reduce_workspace_bytes(n), o1.device)
    call("ssv_negdot_pair_fwd_bwd", n, ptr(o1), ptr(o2), ptr(t1), ptr(t2), float(scale), ptr(loss), ptr(do1), ptr(do2), ptr(ws), ws.numel(), stream())
    return loss, do1, do2


def relic_kl(zi, zj, zo, inv_temp, alpha):
    """ReLIC invariance term on dense [N,D] matrices; returns (alpha * kl as a 0-d tensor, dzi, dzj, dzo)."""
    _lib._dev(zi, zj, zo)
    n, d = zi.shape
    loss = torch.empty((), dtype=torch.float32, device=zi.device)
    dzi, dzj, dzo = torch.empty_like(zi), torch.empty_like(zj), torch.empty_like(zo)
    ws = workspace.get(_lib.load().ssv_relic_kl_workspace_bytes(n), zi.device)
    call("ssv_relic_kl_fwd_bwd", n, d, ptr(zi), ptr(zj), ptr(zo), float(inv_temp), float(alpha), ptr(loss), 0, ptr(dzi), ptr(dzj), ptr(dzo),
         ptr(ws), ws.numel(), stream())
    return loss, dzi, dzj, dzo


def moco_loss(q, k, neg, queue_size, inv_temp):
    """q, k [N,D] (already normalised if the loss normalises), neg [N, ldk] = q bank^T; neg is overwritten by dloss/dneg.
    Returns (loss 0-d, dq_init [N,D])."""
    _lib._dev(q, k, neg)
    n, d = q.shape
    loss = torch.empty((), dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q)
    ws = workspace.get(n * 8, q.device)
    call("ssv_moco_loss_fwd_bwd", n, d, int(queue_size), neg.shape[1], ptr(q), ptr(k), ptr(neg), float(inv_temp), ptr(loss), ptr(dq), ptr(ws), ws.numel(), stream())
    return loss, dq


def queue_push(bank, queue_size, pointer, keys, eps=1e-12):
    """bank[(pointer + i) % queue_size] = normalize(keys[i]); returns the new pointer."""
    _lib._dev(bank, keys)
    invalidate_weight_caches()                        # the bank is a GEMM operand whose transposed image may be cached
    call("ssv_queue_push", int(queue_size), bank.shape[1], ptr(bank), int(pointer), keys.shape[0], ptr(keys), float(eps), stream())
    return (pointer + keys.shape[0]) % queue_size


def queue_push_counted(bank, queue_size, pointer_dev, keys, eps=1e-12):
    """queue_push with the pointer in device memory (an int32 tensor of one element, advanced by the call): what a replayed step needs."""
    _lib._dev(bank, keys, pointer_dev)
    invalidate_weight_caches()
    call("ssv_queue_push_counted", int(queue_size), bank.shape[1], ptr(bank), ptr(pointer_dev), keys.shape[0], ptr(keys), float(eps), stream())


def linear_gelu_fwd(x, w, bias, keep_h=True):
    """(h, gelu(h)) with h = x w^T + bias, both written by one GEMM epilogue.  x [M, C] dense, w [K, C].  ``keep_h`` False (a forward without
    a backward): only gelu(h) is written and h comes back as None."""
    _lib._dev(x, w, bias)
    m, c = x.shape
    w, wshape = _ohwi(w)
    d = conv_desc((m, 1, 1, c), wshape, 1, 0, w=w)
    act = _empty((m, wshape[0]), x)
    h = torch.empty_like(act) if keep_h else None
    call("ssv_linear_gelu_fwd", C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(h), ptr(act), stream())
    return h, act


LINEAR_GELUGRAD_ON_FWD = os.environ.get("SSV_NO_GELUGRAD_FWD_KERNEL", "0") != "1"     # diagnostic switch: fc2's data gradient on the dgrad kernel
GELU_DACT_IN_FWD = os.environ.get("SSV_NO_GELU_DACT", "0") != "1"      # diagnostic switch: the forward keeps the pre-activation and the backward epilogue evaluates gelu' (rounds 1-3)


def can_gelu_dact(w1_shape, w2_shape):
    """fc1 [inter, din] / fc2 [dout, inter] shapes for which the forward can write gelu'(h) and the backward multiply by it on the forward kernel."""
    inter, dout = w1_shape[0], w2_shape[0]
    return GELU_DACT_IN_FWD and LINEAR_GELUGRAD_ON_FWD and dout % 32 == 0 and inter % 4 == 0 and inter >= 128


def linear_gelu_fwd_dact(x, w, bias):
    """(gelu'(h), gelu(h)) with h = x w^T + bias, both written by one GEMM epilogue: the derivative is taken where the cdf is at hand."""
    _lib._dev(x, w, bias)
    m, c = x.shape
    w, wshape = _ohwi(w)
    d = conv_desc((m, 1, 1, c), wshape, 1, 0, w=w)
    act = _empty((m, wshape[0]), x)
    dact = torch.empty_like(act)
    call("ssv_linear_gelu_fwd_dact", C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(dact), ptr(act), stream())
    return dact, act


def linear_dgrad_mul(dy, w, dact, addend=None, out=None):
    """dh = (dy w) * dact (+ addend) with dact = gelu'(h) from linear_gelu_fwd_dact: the forward kernel on the transposed weights."""
    _lib._dev(dy, w, dact, addend)
    m = dy.shape[0]
    w, wshape = _ohwi(w)
    k, c = wshape[0], wshape[1]
    dh = out if out is not None else torch.empty_like(dact)
    wt = _transposed_filter(w, wshape)
    dt = conv_desc((m, 1, 1, k), (c, k, 1, 1), 1, 0, w=wt)
    call("ssv_linear_fwd_mulgrad", C.byref(dt), ptr(dy), ptr(wt), ptr(dact), ptr(addend), ptr(dh), stream())
    return dh


def linear_dgrad_gelu(dy, w, h, addend=None, out=None):
    """dh = (dy w) * gelu'(h) (+ addend): the backward of `gelu(h) -> Linear(w)` down to the pre-activation, in the dgrad epilogue."""
    _lib._dev(dy, w, h, addend)
    m = dy.shape[0]
    w, wshape = _ohwi(w)
    d = conv_desc((m, 1, 1, wshape[1]), wshape, 1, 0)
    dh = out if out is not None else torch.empty_like(h)
    k, c = wshape[0], wshape[1]
    if LINEAR_GELUGRAD_ON_FWD and k % 32 == 0 and c % 4 == 0 and c >= 128:
        # forward kernel on the transposed weights (cached per step like every stride-1 data gradient's): both operands k-contiguous rows
        wt = _transposed_filter(w, wshape)
        dt = conv_desc((m, 1, 1, k), (c, k, 1, 1), 1, 0, w=wt)
        call("ssv_linear_fwd_gelugrad", C.byref(dt), ptr(dy), ptr(wt), ptr(h), ptr(addend), ptr(dh), stream())
        return dh
    call("ssv_conv2d_dgrad_gelu", C.byref(d), ptr(dy), ptr(w), ptr(h), ptr(addend), ptr(dh), stream())
    return dh


def group_expand(wg, groups):
    """Grouped filter bank [K, C/groups, R, S] (OHWI memory) -> dense block-diagonal [K, C, R, S] (OHWI memory)."""
    _lib._dev(wg)
    wg, (k, cg, r, s_) = _ohwi(wg)
    wd = torch.empty((k, cg * groups, r, s_), dtype=torch.float32, device=wg.device).contiguous(memory_format=torch.channels_last)
    call("ssv_group_expand", k, r, s_, cg, groups, ptr(wg), ptr(wd), stream())
    return wd


def group_extract(dwd, dwg, groups, accumulate=True):
    """dwg (+)= the block-diagonal part of the dense weight gradient dwd."""
    _lib._dev(dwd, dwg)
    _, (k, cg, r, s_) = _ohwi(dwg)
    call("ssv_group_extract", k, r, s_, cg, groups, ptr(dwd), ptr(dwg), int(accumulate), stream())
    return dwg


def pad_channels(t, channels):
    """Zero-pad the LAST axis of a dense tensor to `channels` (e.g. an NHWC image batch or an OHWI filter from 3 to 4 channels)."""
    _lib._dev(t)
    cin = t.shape[-1]
    out = torch.empty(t.shape[:-1] + (channels,), dtype=torch.float32, device=t.device)
    call("ssv_pad_channels", t.numel() // cin, cin, channels, ptr(t), ptr(out), 0, stream())
    return out


def unpad_channels(t, out, accumulate=True):
    """out (+)= the first out.shape[-1] channels of t (both dense, channel axis last)."""
    _lib._dev(t, out)
    call("ssv_pad_channels", t.numel() // t.shape[-1], t.shape[-1], out.shape[-1], ptr(t), ptr(out), int(accumulate), stream())
    return out


# ------------------------------------------------------------------------------------------- Winograd F(2x2, 3x3)
def _wino_filter(w, wshape, transposed=False):
    """U = G g G^T of the filter [K][3][3][C] - or, ``transposed``, of the rotated filter with the channel roles swapped (data gradient)."""
    key = (w.data_ptr(), stream(), wshape, transposed)
    hit = _WINO_U.get(key)
    if hit is not None:
        return hit[1]
    k, c, _, _ = wshape
    src, kk, cc = (w, k, c) if not transposed else (_transposed_filter(w, wshape), c, k)
    u = torch.empty((16, kk, cc), dtype=torch.float32, device=w.device)
    call("ssv_wino_filter_transform", kk, cc, ptr(src), ptr(u), stream())
    _WINO_U[key] = (w.untyped_storage(), u)
    return u


WINOGRAD = os.environ.get("SSV_NO_WINOGRAD", "0") != "1"          # diagnostic switch: every 3x3 convolution on the direct implicit-GEMM kernels
# Measured at bs 512 (tools/probe_winograd.py, profiles/r03_probe_winograd.txt): 14x14 x 256 channels forward 1.50x, data gradient 1.62x, weight
# gradient 1.97x; 7x7 x 512: 1.76 / 1.73 / 1.72; 28x28 x 128: 1.13 / 1.10 / 1.54; 56x56 x 64: 0.79 / 0.71 (the transforms move 9x the tensors
# and that layer is HBM-heavy already) - so: at least 128 channels on both sides.
WINOGRAD_MIN_CHANNELS = int(os.environ.get("SSV_WINOGRAD_MIN_CHANNELS", "128"))
# Round 5: that verdict was F(2x2)'s.  With all three products on F(4x4) (2.25x instead of 4x transformed tensors, no second transformed input, one pass over dY) the
# 64-channel 3x3 layers of layer1 (56x56) win too: same-box A/B 215.9 -> 210.7 ms per step (profiles/r05_probe_wino64_step_ab.txt).  Narrower layers than
# WINOGRAD_MIN_CHANNELS therefore take Winograd from this width on - but only where F(4x4) is what runs (`_use_wino44`), never F(2x2).
# Round 6: on the bf16x3 arithmetic the direct product is 1.4x cheaper and the verdict flips back - the three 64-channel layers on the direct kernels: +0.5 % images/s
# and 707 -> 654 GB of HBM traffic per step (same box, alternating, profiles/r06_probe_wino_floor_traffic.txt; the 128-channel layers direct as well: 625 GB but
# -1.7 %).  Default therefore by arithmetic: 128 on bf16x3, 64 on fp32 MFMA; SSV_WINOGRAD44_MIN_CHANNELS=<n> overrides both.
WINOGRAD44_MIN_CHANNELS = int(os.environ["SSV_WINOGRAD44_MIN_CHANNELS"]) if os.environ.get("SSV_WINOGRAD44_MIN_CHANNELS") else None


def _wino44_min_channels():
    return WINOGRAD44_MIN_CHANNELS if WINOGRAD44_MIN_CHANNELS is not None else (128 if ARITHMETIC == "bf16x3" else 64)


WINOGRAD_MIN_TILES = int(os.environ.get("SSV_WINOGRAD_MIN_TILES", "256"))     # below that the three launches cost more than they save


def _lanes_ok(ch):
    return ch % 32 == 0 and ((ch // 4 <= 256 and 256 % (ch // 4) == 0) or (ch // 4) % 256 == 0)


def use_winograd(wshape, stride, pad, x_shape, want_stats):
    """Does this convolution run through F(2x2, 3x3)?  3x3 / stride 1 / padding 1, channel counts the transforms' lane mapping takes, enough
    tiles, every operand slice below the per-launch limit, and - when the statistics epilogue is wanted - a map that partitions evenly."""
    k, c, r, s_ = wshape
    if not (WINOGRAD and r == 3 and s_ == 3 and stride == 1 and pad == 1 and _lanes_ok(c) and _lanes_ok(k)):
        return False
    n, h, w_ = x_shape[0], x_shape[1], x_shape[2]
    if min(c, k) < WINOGRAD_MIN_CHANNELS:          # narrow layers: only where every product runs F(4x4) (forward / data / weight gradient share the 0.75 ratio rule)
        # (the forward's own ratio too: a narrow layer admitted on the data gradient's ratio alone would run its forward on F(2x2) when SSV_WINOGRAD44_FWD_RATIO is lowered)
        if not (min(c, k) >= _wino44_min_channels() and WINOGRAD44_WGRAD
                and _use_wino44(n, h, w_, c, k, min(WINOGRAD44_MAX_RATIO_FWD_NO_V2, WINOGRAD44_MAX_RATIO_DGRAD))):
            return False
    t = n * ((h + 1) // 2) * ((w_ + 1) // 2)
    if t < WINOGRAD_MIN_TILES or t * max(c, k) >= _MAX_ELEMS:
        return False
    return not want_stats or int(_lib.load().ssv_wino_stats_rows_per_group(n, h, w_)) > 0


# F(4x4, 3x3) for the forward and the data gradient of the Winograd layers (csrc/winograd44.hip; the weight gradient stays on F(2x2), whose operand the F(4x4) input
# transform leaves beside its own).  SSV_WINOGRAD44=0|1 overrides the shipped default.
WINOGRAD44 = os.environ.get("SSV_WINOGRAD44", "1") == "1"
# Which product takes F(4x4): by the transformed-domain work it leaves, 36 positions x tiles of 4 against 16 x tiles of 2 (0.5625 on maps that tile evenly - 28x28 -
# and on 7x7, where both tilings cover 8; 0.735 on 14x14, which F(2x2) tiles exactly and F(4x4) covers with 16).  The forward pays for writing both transformed
# inputs (its own and the weight gradient's), so it switches only when the work halves; the data gradient already wins at 0.735 (tools/probe/wino44_stages.py,
# profiles/r04_probe_wino44_stages.txt: 14x14 forward 1.07x, data gradient 1.28x; go / no-go bar 1.25x).
WINOGRAD44_MAX_RATIO_FWD, WINOGRAD44_MAX_RATIO_DGRAD = 0.6, 0.75
# Round 5: with the weight gradient on F(4x4) too (WINOGRAD44_WGRAD) the forward no longer writes the second, F(2x2) transformed input - its input transform drops from
# 0.209 to 0.133 ms on 14x14 x 256 - and the weight gradient it feeds is 0.485 instead of 0.548 ms: the forward then switches at the data gradient's ratio
# (14x14: 0.70 -> 0.57 ms forward; profiles/r05_probe_winograd44_wgrad.txt, r04_probe_wino44_stages.txt).
WINOGRAD44_MAX_RATIO_FWD_NO_V2 = float(os.environ.get("SSV_WINOGRAD44_FWD_RATIO", "0.75"))
# ... and only up to the contraction length the numerics bar was measured for: the transformed-domain sums run over the channels and F(4x4)'s error grows with
# their count like any fp32 sum, but from a 3x higher base - 128 / 256 / 512 channels: 2.97x / 2.84x / 2.85x the direct kernel's error against fp64 (bar: 3x),
# 1024 channels (wide_resnet's layer4): 3.8x.  Wider layers stay on F(2x2).
WINOGRAD44_MAX_CHANNELS = 512
# ... and only for launches with enough tiles: 36 small GEMMs of a few row tiles each buy nothing over 16 (B = 64: 7x7 x 512, 256 tiles: 0.109 vs 0.116 ms) while
# the larger rounding error stays - small batches keep F(2x2)
WINOGRAD44_MIN_TILES = int(os.environ.get("SSV_WINOGRAD44_MIN_TILES", "1024"))
# The weight gradient of a layer whose FORWARD ran F(4x4) takes F(4x4) too (round 5): its operand is the V the forward's input transform wrote anyway (no second,
# F(2x2) transformed input), 36 products over a quarter of the tiles, a 2.25x dY transform.  Error against fp64 1.1 - 1.6e-6 relative (bar 2e-6: a weight gradient's
# error never crosses a ReLU gate).  SSV_WINOGRAD44_WGRAD=0: those layers keep the F(2x2) operand and weight gradient (round 4's selection).
WINOGRAD44_WGRAD = os.environ.get("SSV_WINOGRAD44_WGRAD", "1") == "1"
# ... with BLOCKED accumulation of its 36 transformed-domain sums: fp32 chains of at most this many tiles, the chunks folded in fp64 (ssv_gemm_batched_wgrad_blocked);
# 0 = the plain split (diagnostic: how the error grows with the chain length)
WINOGRAD44_WGRAD_CHUNK = int(os.environ.get("SSV_WINOGRAD44_WGRAD_CHUNK", "512"))
# ... or / and inside the kernel: the MFMA accumulators flushed into a second register set every 128 tiles (0 = off)
WINOGRAD44_WGRAD_FLUSH = int(os.environ.get("SSV_WINOGRAD44_WGRAD_FLUSH", "128"))
WINOGRAD44_WGRAD_FLUSH_BF16X3 = 0
# The backward of such a layer reads its output gradient ONCE: the weight gradient's dY transform also writes the data gradient's transformed input
# (ssv_wino44_dy_transform_both; the data gradient that follows picks it up), and behind a fused input chain the output gradient is never written at all - the
# BatchNorm backward hands over (g, x, coefficients) as a LazyGrad and the transform forms it on load (`wino44_lazy_dy_ok`).  SSV_WINOGRAD44_DY_BOTH=0: separate passes.
WINOGRAD44_DY_BOTH = os.environ.get("SSV_WINOGRAD44_DY_BOTH", "1") == "1"


def wino44_lazy_dy_ok(y, w_shape, x_shape):
    """May the BatchNorm behind this convolution output hand its backward over as a LazyGrad?  Only when forward, data gradient and weight gradient all run F(4x4):
    the output carries F(4x4)'s own transformed input (the weight gradient will take the (g, x) operand) and the data gradient's dispatch rule holds for dy's shape
    (= the output's: 3x3 / stride 1 / padding 1)."""
    v = getattr(y, "_wino_v", None)
    k, c, _, _ = w_shape
    return (WINOGRAD44_DY_BOTH and WINOGRAD44_WGRAD and v is not None and v.shape[0] == 36
            and _use_wino44(y.shape[0], y.shape[1], y.shape[2], c, k, WINOGRAD44_MAX_RATIO_DGRAD))


# Which Winograd form each product took (launch counts by name), recorded while DISPATCH is a dict: bench.py's parity gate prints it so that the line
# shows the gate ran the kernel selection of the timed batch-512 step (`large_batch_dispatch`)
DISPATCH = None


def _note(kind):
    if DISPATCH is not None:
        DISPATCH[kind] = DISPATCH.get(kind, 0) + 1


class large_batch_dispatch:
    """Run a SMALL batch through the kernel selection of a large one: the tile-count floors of the two Winograd forms are lifted, so every 3x3 product
    that takes F(4x4) / F(2x2) at batch 512 takes it here too (the ratio / channel rules are batch-independent).  Test and gate plumbing only."""

    def __enter__(self):
        global WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES, DISPATCH
        self.prev = (WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES, DISPATCH)
        WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES, DISPATCH = 0, 0, {}
        self.log = DISPATCH
        return self

    def __exit__(self, *exc):
        global WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES, DISPATCH
        WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES, DISPATCH = self.prev
        return False


class graph_dispatch:
    """The kernel selection for a step that is being captured into a HIP graph (graph.StepGraph) on SMALL images: without a per-launch host cost the Winograd forms
    pay from far fewer tiles and from 64 channels - their 16 / 36 batched products put 16 / 36 times the workgroups of the direct kernel on maps whose direct
    launch fills a fraction of the chip (resnet18 at 32 x 32, batch 64: layer4's 3x3 is 32 workgroups walking 144 k-tiles each).  Measured through the graph,
    ms per step at batch 64 / 512 (profiles/r05_cifar_wino_floors.txt): default floors 6.79 / 9.47, these 5.17 / 9.31; launched kernel by kernel the same
    selection is SLOWER (10.6 vs 9.2 at batch 64: more launches) - hence only under capture."""
    CHANNELS, TILES, TILES44 = 64, 64, 256

    def __enter__(self):
        global WINOGRAD_MIN_CHANNELS, WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES
        self.prev = (WINOGRAD_MIN_CHANNELS, WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES)
        WINOGRAD_MIN_CHANNELS, WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES = (min(self.prev[0], self.CHANNELS), min(self.prev[1], self.TILES), min(self.prev[2], self.TILES44))
        return self

    def __exit__(self, *exc):
        global WINOGRAD_MIN_CHANNELS, WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES
        WINOGRAD_MIN_CHANNELS, WINOGRAD_MIN_TILES, WINOGRAD44_MIN_TILES = self.prev
        return False


def _wino44_ratio(h, w_):
    return (36.0 * ((h + 3) // 4) * ((w_ + 3) // 4)) / (16.0 * ((h + 1) // 2) * ((w_ + 1) // 2))


def _use_wino44(n, h, w_, c, k, max_ratio):
    return (WINOGRAD44 and max(c, k) <= WINOGRAD44_MAX_CHANNELS and _wino44_ratio(h, w_) <= max_ratio
            and n * ((h + 3) // 4) * ((w_ + 3) // 4) >= WINOGRAD44_MIN_TILES)


def _wino44_filter(w, wshape, transposed=False):
    """U = G g G^T (6x6 positions) of the filter [K][3][3][C] - or, ``transposed``, of the rotated filter with the channel roles swapped (data gradient)."""
    key = (w.data_ptr(), stream(), wshape, transposed, 44)
    hit = _WINO_U.get(key)
    if hit is not None:
        return hit[1]
    k, c, _, _ = wshape
    src, kk, cc = (w, k, c) if not transposed else (_transposed_filter(w, wshape), c, k)
    u = torch.empty((36, kk, cc), dtype=torch.float32, device=w.device)
    call("ssv_wino44_filter_transform", kk, cc, ptr(src), ptr(u), stream())
    _WINO_U[key] = (w.untyped_storage(), u)
    return u


def _gemm_batched(nb, t, c, k, a, u, m):
    """m[b] = a[b] . u[b]^T for the nb transformed-domain positions (a [nb][t][c], u [nb][k][c], m [nb][t][k]); in the bf16x3 arithmetic on the planes of u."""
    if ARITHMETIC == "bf16x3" and c % 32 == 0 and k % 4 == 0:
        _note("gemm_batched_bf16x3")
        call("ssv_gemm_batched_split", nb, t, c, k, ptr(a), ptr(_planes(u)), ptr(m), None, None, stream())
    else:
        call("ssv_gemm_batched", nb, t, c, k, ptr(a), ptr(u), ptr(m), stream())


def _gemm_batched_wgrad(nb, t, c, k, v, dm, du, chunk=0, flush=0):
    """du[b] = dm[b]^T . v[b] over the t tiles; ``chunk`` / ``flush``: blocked accumulation (ssv_gemm_batched_wgrad_blocked)."""
    lib = _lib.load()
    if ARITHMETIC == "bf16x3" and c % 4 == 0 and k % 4 == 0:
        _note("gemm_batched_wgrad_bf16x3")
        ws = workspace.get(lib.ssv_gemm_batched_wgrad_blocked_workspace_bytes(nb, t, c, k, chunk), v.device)
        call("ssv_gemm_batched_wgrad_split", nb, t, c, k, ptr(v), ptr(dm), ptr(du), chunk, flush, ptr(ws), ws.numel(), stream())
    elif chunk > 0 or flush > 0:
        ws = workspace.get(lib.ssv_gemm_batched_wgrad_blocked_workspace_bytes(nb, t, c, k, chunk), v.device)
        call("ssv_gemm_batched_wgrad_blocked", nb, t, c, k, ptr(v), ptr(dm), ptr(du), chunk, flush, ptr(ws), ws.numel(), stream())
    else:
        ws = workspace.get(lib.ssv_gemm_batched_wgrad_workspace_bytes(nb, t, c, k), v.device)
        call("ssv_gemm_batched_wgrad", nb, t, c, k, ptr(v), ptr(dm), ptr(du), ptr(ws), ws.numel(), stream())


def wino44_conv2d_fwd(x, w, in_affine=None, want_stats=False, keep_v=False):
    """wino_conv2d_fwd through F(4x4, 3x3): 36 transformed-domain GEMMs over a quarter of the tiles.  ``keep_v``: the weight gradient's operand comes back as
    the third result - the transformed input V [36][T][C] itself (WINOGRAD44_WGRAD: the weight gradient runs F(4x4) too), or the F(2x2) transformed input
    [16][T2][C] the input transform then also leaves (wino_conv2d_wgrad tells them apart by the leading dimension).  Statistics partials: one per row of
    tiles (H % 4 == 0) or per image."""
    _note("wino44_fwd")
    _lib._dev(x, w)
    w, wshape = _ohwi(w)
    n, h, w_, c = x.shape
    k = wshape[0]
    lib = _lib.load()
    t = int(lib.ssv_wino44_tiles(n, h, w_))
    u = _wino44_filter(w, wshape)
    sc, sh = in_affine if in_affine is not None else (None, None)
    v = torch.empty((36, t, c), dtype=torch.float32, device=x.device)
    v2 = torch.empty((16, int(lib.ssv_wino_tiles(n, h, w_)), c), dtype=torch.float32, device=x.device) if (keep_v and not WINOGRAD44_WGRAD) else None
    call("ssv_wino44_input_transform", n, h, w_, c, ptr(x), ptr(sc), ptr(sh), ptr(v), ptr(v2), stream())
    if keep_v and WINOGRAD44_WGRAD:
        v2 = v
    m = torch.empty((36, t, k), dtype=torch.float32, device=x.device)
    _gemm_batched(36, t, c, k, v, u, m)
    y = _empty((n, h, w_, k), x)
    part = None
    if want_stats:
        part = _empty((2, int(lib.ssv_wino44_groups(n, h, w_, 1)), k), x)
    call("ssv_wino44_output_transform", n, h, w_, k, ptr(m), ptr(y), None if part is None else ptr(part[0]), None if part is None else ptr(part[1]), None, stream())
    rpg = int(lib.ssv_wino44_stats_rows_per_group(n, h, w_)) if want_stats else 0
    return y, (None if part is None else (part[0], part[1], rpg)), v2


def wino44_conv2d_dgrad(dy, w, gate=None):
    """wino_conv2d_dgrad through F(4x4, 3x3); the gate's partial sums come one per row of tiles.  ``dy`` may be the LazyGrad / tensor whose transformed input
    the weight gradient's pass already wrote (`wino44_conv2d_wgrad`): the input transform is then skipped."""
    _note("wino44_dgrad")
    vd = None
    if isinstance(dy, LazyGrad):
        vd, dy.wino_vd, dy = dy.wino_vd, None, dy.g
        if vd is None:
            raise _lib.SsvError("wino44_conv2d_dgrad: a LazyGrad without the transformed input the weight gradient leaves")
    else:
        vd = dy.__dict__.pop("_wino44_vd", None)
    _lib._dev(dy, w)
    w, wshape = _ohwi(w)
    n, h, w_, k = dy.shape
    c = wshape[1]
    lib = _lib.load()
    t = int(lib.ssv_wino44_tiles(n, h, w_))
    u = _wino44_filter(w, wshape, transposed=True)                # [36][C][K]
    if vd is not None and tuple(vd.shape) == (36, t, k):
        v = vd
    else:
        v = torch.empty((36, t, k), dtype=torch.float32, device=dy.device)
        call("ssv_wino44_input_transform", n, h, w_, k, ptr(dy), None, None, ptr(v), None, stream())
    m = torch.empty((36, t, c), dtype=torch.float32, device=dy.device)
    _gemm_batched(36, t, k, c, v, u, m)
    dx = _empty((n, h, w_, c), dy)
    if gate is not None:
        groups = int(lib.ssv_wino44_groups(n, h, w_, 0))
        st, part = _gate_struct(gate, groups, c, dy)
        call("ssv_wino44_output_transform", n, h, w_, c, ptr(m), ptr(dx), None, None, C.byref(st), stream())
        dx._gate_partials = (part[0], part[1], groups)
    else:
        call("ssv_wino44_output_transform", n, h, w_, c, ptr(m), ptr(dx), None, None, None, stream())
    return dx


def wino_conv2d_fwd(x, w, in_affine=None, want_stats=False, keep_v=False):
    """y = conv3x3(act(x), w) (stride 1, padding 1) through F(2x2, 3x3).  Returns (y, (pmean, pm2) | None, V | None): the statistics
    partials are one per 16 tiles = 64 output rows (needs even H, W); V is the transformed input, kept for the weight gradient."""
    if _use_wino44(x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0], WINOGRAD44_MAX_RATIO_FWD_NO_V2 if WINOGRAD44_WGRAD else WINOGRAD44_MAX_RATIO_FWD):
        return wino44_conv2d_fwd(x, w, in_affine, want_stats, keep_v)
    _note("wino22_fwd")
    _lib._dev(x, w)
    w, wshape = _ohwi(w)
    n, h, w_, c = x.shape
    k = wshape[0]
    lib = _lib.load()
    t = int(lib.ssv_wino_tiles(n, h, w_))
    u = _wino_filter(w, wshape)
    sc, sh = in_affine if in_affine is not None else (None, None)
    v = torch.empty((16, t, c), dtype=torch.float32, device=x.device)
    call("ssv_wino_input_transform", n, h, w_, c, ptr(x), ptr(sc), ptr(sh), ptr(v), stream())
    m = torch.empty((16, t, k), dtype=torch.float32, device=x.device)
    _gemm_batched(16, t, c, k, v, u, m)
    y = _empty((n, h, w_, k), x)
    part = None
    if want_stats:
        groups = int(lib.ssv_wino_groups(n, h, w_))
        part = _empty((2, groups, k), x)
    call("ssv_wino_output_transform", n, h, w_, k, ptr(m), ptr(y), None if part is None else ptr(part[0]), None if part is None else ptr(part[1]), None, stream())
    rpg = int(lib.ssv_wino_stats_rows_per_group(n, h, w_)) if want_stats else 0
    return y, (None if part is None else (part[0], part[1], rpg)), (v if keep_v else None)


def wino_conv2d_dgrad(dy, w, gate=None):
    """dx = conv3x3 data gradient (stride 1, padding 1) through F(2x2, 3x3) on the transposed, rotated filter; ``gate`` (BnGateCtx without
    a second target): dx is gated and its partial sums come back as ``dx._gate_partials`` like conv2d_dgrad's."""
    if _use_wino44(dy.shape[0], dy.shape[1], dy.shape[2], w.shape[1], dy.shape[3], WINOGRAD44_MAX_RATIO_DGRAD):
        return wino44_conv2d_dgrad(dy, w, gate)
    _note("wino22_dgrad")
    _lib._dev(dy, w)
    w, wshape = _ohwi(w)
    n, h, w_, k = dy.shape
    c = wshape[1]
    lib = _lib.load()
    t = int(lib.ssv_wino_tiles(n, h, w_))
    u = _wino_filter(w, wshape, transposed=True)                  # [16][C][K]
    v = torch.empty((16, t, k), dtype=torch.float32, device=dy.device)
    call("ssv_wino_input_transform", n, h, w_, k, ptr(dy), None, None, ptr(v), stream())
    m = torch.empty((16, t, c), dtype=torch.float32, device=dy.device)
    _gemm_batched(16, t, k, c, v, u, m)
    dx = _empty((n, h, w_, c), dy)
    if gate is not None:
        groups = int(lib.ssv_wino_groups(n, h, w_))
        st, part = _gate_struct(gate, groups, c, dy)
        call("ssv_wino_output_transform", n, h, w_, c, ptr(m), ptr(dx), None, None, C.byref(st), stream())
        dx._gate_partials = (part[0], part[1], groups)
    else:
        call("ssv_wino_output_transform", n, h, w_, c, ptr(m), ptr(dx), None, None, None, stream())
    return dx


def wino_conv2d_wgrad(v, dy, w_like, dw, accumulate=True, dgrad_follows=None):
    """dw (+)= weight gradient of the 3x3 convolution whose transformed input V was kept by wino_conv2d_fwd ([16][T2][C]: F(2x2); [36][T][C]: F(4x4))."""
    if v.shape[0] == 36:
        return wino44_conv2d_wgrad(v, dy, w_like, dw, accumulate, dgrad_follows=dgrad_follows)
    _note("wino22_wgrad")
    _lib._dev(v, dy, dw)
    _, wshape = _ohwi(w_like)
    n, h, w_, k = dy.shape
    c = wshape[1]
    lib = _lib.load()
    t = int(lib.ssv_wino_tiles(n, h, w_))
    dm = torch.empty((16, t, k), dtype=torch.float32, device=dy.device)
    call("ssv_wino_dy_transform", n, h, w_, k, ptr(dy), ptr(dm), stream())
    du = torch.empty((16, k, c), dtype=torch.float32, device=dy.device)
    _gemm_batched_wgrad(16, t, c, k, v, dm, du)
    call("ssv_wino_filter_grad", k, c, ptr(du), ptr(dw), int(accumulate), stream())
    return dw


def wino44_conv2d_wgrad(v, dy, w_like, dw, accumulate=True, dgrad_follows=None):
    """wino_conv2d_wgrad through F(4x4, 3x3): dM = A dY A^T, 36 products dU_p = dM_p^T V_p over the tiles, dw (+)= G^T dU G."""
    _note("wino44_wgrad")
    _lib._dev(v, dy.g if isinstance(dy, LazyGrad) else dy, dw)
    _, wshape = _ohwi(w_like)
    n, h, w_, k = dy.shape
    c = wshape[1]
    lib = _lib.load()
    t = int(lib.ssv_wino44_tiles(n, h, w_))
    if tuple(v.shape) != (36, t, c):
        raise _lib.SsvError(f"wino44_conv2d_wgrad: transformed input {tuple(v.shape)} does not belong to a {n}x{h}x{w_}x{c} activation")
    lazy = dy if isinstance(dy, LazyGrad) else None
    g = lazy.g if lazy is not None else dy
    dm = torch.empty((36, t, k), dtype=torch.float32, device=g.device)
    # the data gradient of the same layer follows (nn.conv: weight gradient first): when it will run F(4x4) too, its transformed input comes out of this pass
    both = WINOGRAD44_DY_BOTH and (lazy is not None or (dgrad_follows is not False and _use_wino44(n, h, w_, c, k, WINOGRAD44_MAX_RATIO_DGRAD)))
    if lazy is not None and not both:
        raise _lib.SsvError("wino44_conv2d_wgrad: a LazyGrad needs the one-pass transform (ops.WINOGRAD44_DY_BOTH was switched off after the forward marked this layer)")
    if both:
        vd = torch.empty((36, t, k), dtype=torch.float32, device=g.device)
        dyin = _dyin_struct(lazy, 0, n) if lazy is not None else None
        _note("wino44_dy_both_formed_on_load" if lazy is not None else "wino44_dy_both")
        call("ssv_wino44_dy_transform_both", n, h, w_, k, ptr(g), None if dyin is None else C.byref(dyin), ptr(vd), ptr(dm), stream())
        if lazy is not None:
            lazy.wino_vd = vd
        else:
            dy._wino44_vd = vd
    else:
        call("ssv_wino44_dy_transform", n, h, w_, k, ptr(g), ptr(dm), stream())
    dy = g
    du = torch.empty((36, k, c), dtype=torch.float32, device=dy.device)
    if ARITHMETIC == "bf16x3":
        # the bf16 instruction folds 32 products per accumulator rounding (the fp32 one: 2), so a chain's rounding error is that of one 16x shorter: the register-level
        # flush is not needed for the 2e-6 bar (tests/test_gpu_winograd44.py measures it), the chunked fp64 fold stays
        _gemm_batched_wgrad(36, t, c, k, v, dm, du, WINOGRAD44_WGRAD_CHUNK, WINOGRAD44_WGRAD_FLUSH_BF16X3)
    else:
        _gemm_batched_wgrad(36, t, c, k, v, dm, du, WINOGRAD44_WGRAD_CHUNK, WINOGRAD44_WGRAD_FLUSH)
    call("ssv_wino44_filter_grad", k, c, ptr(du), ptr(dw), int(accumulate), stream())
    return dw


# ------------------------------------------------------------------------------------------- the 3-channel image stem, row-taps form
def can_row_stem(w_shape):
    k, c, r, s_ = w_shape
    return c == 3 and 3 * s_ <= 24 and s_ <= 8 and k % 4 == 0


def stem_weight_rows(w):
    """[K,3,R,S] filter (OHWI memory) -> [K*R][24]: its rows of 3 S floats, (s, c) order, zero-padded to 24."""
    w, (k, c, r, s_) = _ohwi(w)
    return pad_channels(w.permute(0, 2, 3, 1).reshape(k * r, s_ * c), 24)


def stem_conv_fwd(x, wrows, w_shape, stride, pad, want_stats=False):
    """y = conv(x, w) for a 3-channel NHWC image batch, x UNPADDED: returns (y, (pmean, pm2) | None)."""
    _lib._dev(x, wrows)
    d = conv_desc(x.shape, w_shape, stride, pad)
    y = _empty((d.N, d.Ho, d.Wo, d.K), x)
    lib = _lib.load()
    # statistics partials: one per 64 output pixels, or per whole output rows on the rows-in-LDS kernel (the same for every chunk: it depends on the map, not the batch)
    rpg = int(lib.ssv_stem_conv_fwd_stats_rows_per_group(C.byref(d)))
    m = d.N * d.Ho * d.Wo
    part = _empty((2, -(-m // rpg), d.K), x) if want_stats else None
    g0 = 0
    for n0, n1 in _batch_chunks(d.N, (d.H * d.W * d.C, d.Ho * d.Wo * d.K), rows_per_sample=d.Ho * d.Wo if want_stats else None):
        dc = conv_desc((n1 - n0,) + tuple(x.shape[1:]), w_shape, stride, pad)
        if int(lib.ssv_stem_conv_fwd_stats_rows_per_group(C.byref(dc))) != rpg:
            raise _lib.SsvError("ssv_stem_conv_fwd: statistics group size changed between batch chunks")
        gc = -(-(n1 - n0) * d.Ho * d.Wo // rpg)
        call("ssv_stem_conv_fwd", C.byref(dc), ptr(x[n0:n1]), ptr(wrows), ptr(y[n0:n1]),
             ptr(part[0][g0:g0 + gc]) if want_stats else None, ptr(part[1][g0:g0 + gc]) if want_stats else None, stream())
        g0 += gc
    if part is None:
        return y, None
    return y, ((part[0], part[1]) if rpg == 64 else (part[0], part[1], rpg))


def stem_conv_wgrad(x, dy, w_shape, stride, pad):
    """Weight gradient in the row-taps layout: [K*R][24] (columns >= 3 S are zero)."""
    _lib._dev(x, dy)
    k, c, r, s_ = w_shape
    lib = _lib.load()
    out = None
    for n0, n1 in _batch_chunks(x.shape[0], (x[0].numel(), dy[0].numel())):
        d = conv_desc(x[n0:n1].shape, w_shape, stride, pad)
        ws = workspace.get(lib.ssv_stem_conv_wgrad_workspace_bytes(C.byref(d)), x.device)
        dwr = torch.empty((k * r, 24), dtype=torch.float32, device=x.device)
        call("ssv_stem_conv_wgrad", C.byref(d), ptr(x[n0:n1]), ptr(dy[n0:n1]), ptr(dwr), ptr(ws), ws.numel(), stream())
        out = dwr if out is None else add_(out, dwr)
    return out
