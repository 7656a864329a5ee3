"""Tensor-level wrappers over the C ABI.  Activations are NHWC fp32 device tensors, filters are
[O,I,H,W] tensors in channels_last memory format (= OHWI in memory).  Every function only enqueues
kernels on the current HIP stream; torch provides allocation, nothing else."""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, call, ptr, stream, workspace

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def conv_desc(x_shape, w_shape, stride, pad):
    n, h, w, c = x_shape
    k, ci, r, s = w_shape
    if ci != c:
        raise _lib.SsvError(f"conv: input has {c} channels, filter expects {ci}")
    ho, wo = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1
    return ConvDesc(n, h, w, c, k, r, s, stride, pad, ho, wo)


def _ohwi(w):
    """The filter must be OHWI in memory: a channels_last [O,I,H,W] tensor (or any [O,I,1,1]/2-D matrix)."""
    if w.dim() == 2:
        if not w.is_contiguous():
            raise _lib.SsvError("linear weight must be contiguous")
        return w, (w.shape[0], w.shape[1], 1, 1)
    if not w.is_contiguous(memory_format=torch.channels_last):
        raise _lib.SsvError("conv filter must be in channels_last (OHWI) memory format")
    return w, tuple(w.shape)


def conv2d_fwd(x, w, stride=1, pad=0, bias=None, addend=None):
    _lib._dev(x, w, bias, addend)
    w, wshape = _ohwi(w)
    d = conv_desc(x.shape, wshape, stride, pad)
    y = _empty((d.N, d.Ho, d.Wo, d.K), x)
    call("ssv_conv2d_fwd", C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(addend), ptr(y), stream())
    return y


def conv2d_dgrad(dy, w, x_shape, stride=1, pad=0, addend=None, out=None):
    _lib._dev(dy, w, addend)
    w, wshape = _ohwi(w)
    d = conv_desc(x_shape, wshape, stride, pad)
    dx = out if out is not None else _empty(tuple(x_shape), dy)
    call("ssv_conv2d_dgrad", C.byref(d), ptr(dy), ptr(w), ptr(addend), ptr(dx), stream())
    return dx


def conv2d_wgrad(x, dy, w_like, dw, stride=1, pad=0, accumulate=True):
    """dw (+)= wgrad.  ``dw`` has the memory layout of ``w_like`` (OHWI)."""
    _lib._dev(x, dy, dw)
    _, wshape = _ohwi(w_like)
    d = conv_desc(x.shape, wshape, stride, pad)
    nbytes = _lib.load().ssv_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = workspace.get(nbytes, x.device)
    call("ssv_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), int(accumulate), ptr(ws), ws.numel(), stream())
    return dw


def _rows(x):
    c = x.shape[-1]
    return x.numel() // c, c


def bn_train_fwd(x, gamma, beta, running_mean, running_var, nbt, relu=False, residual=None,
                 eps=BN_EPS, momentum=BN_MOMENTUM, want_mask=False, skip_mask=False):
    """Returns (y, mean, invstd[, relu_mask]).  relu_mask: uint8, one byte per four channels, for the backward."""
    _lib._dev(x, gamma, beta, residual)
    m, c = _rows(x)
    y = torch.empty_like(x)
    mean, invstd = _empty((c,), x), _empty((c,), x)
    mask = torch.empty((m * c // 4,), dtype=torch.uint8, device=x.device) if (want_mask and relu and not skip_mask) else None
    ws = workspace.get(_lib.load().ssv_bn_workspace_bytes(m, c), x.device)
    call("ssv_bn_train_fwd", m, c, ptr(x), ptr(gamma), ptr(beta), ptr(residual), int(relu), eps, momentum,
         ptr(running_mean), ptr(running_var), ptr(nbt), ptr(y), ptr(mask), ptr(mean), ptr(invstd), ptr(ws), ws.numel(), stream())
    return (y, mean, invstd, mask) if want_mask else (y, mean, invstd)


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


def scale_(x, factor_dev):
    call("ssv_scale", x.numel(), ptr(x), ptr(factor_dev), stream())
    return x


def fill_(x, value):
    call("ssv_fill", x.numel(), ptr(x), float(value), stream())
    return x


def add_(dst, src):
    call("ssv_add", dst.numel(), ptr(dst), ptr(src), stream())
    return dst


def ntxent_fwd(zall, nglob, b, seg0, inv_temp):
    """Row log-sum-exp and positive logit of this rank's 2*b rows against the gathered [2*nglob, ld] matrix."""
    _lib._dev(zall)
    lse = _empty((2 * b,), zall)
    pos = _empty((2 * b,), zall)
    call("ssv_ntxent_fwd", nglob, b, seg0, zall.shape[1], ptr(zall), float(inv_temp), ptr(lse), ptr(pos), stream())
    return lse, pos


def ntxent_loss(lse, pos, scale):
    loss = torch.empty((), dtype=torch.float32, device=lse.device)
    call("ssv_ntxent_loss", lse.numel(), ptr(lse), ptr(pos), float(scale), ptr(loss), stream())
    return loss


def ntxent_bwd(zall, lse_all, nglob, b, seg0, inv_temp, gscale):
    _lib._dev(zall, lse_all)
    dz = _empty((2 * b, zall.shape[1]), zall)
    call("ssv_ntxent_bwd", nglob, b, seg0, zall.shape[1], ptr(zall), ptr(lse_all), float(inv_temp), float(gscale), ptr(dz), stream())
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
    ws = torch.empty(_lib.load().ssv_knn_workspace_bytes(n), dtype=torch.uint8, device=z.device)    # up to ~1 GB: not kept in the training scratch
    call("ssv_knn_label_agreement", n, d, ptr(z), ptr(labels), int(k), ptr(count), ptr(ws), ws.numel(), stream())
    return int(count.item())
