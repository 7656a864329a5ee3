"""Host-side layer engine: a small reverse-mode tape over the HIP ops, bridged into torch.autograd
at MODULE granularity (one autograd node per network call), so the reference's
``loss.backward(); optim.step()`` surface keeps working while every kernel on the path is ours.

Parameters are ordinary ``nn.Parameter``s (state_dict keys/shapes identical to the reference);
parameter gradients are accumulated by the kernels straight into ``p.grad`` (which the optimizer
keeps as views of one flat arena), never through torch ops.
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import SsvError


# ------------------------------------------------------------------------------------------- tape
class Tape:
    """Records (inputs, output, backward closure) per op.  ``backward`` walks it in reverse; a
    closure gets the gradient of its output plus any gradient already accumulated for each of
    its inputs, so producers that can fuse ``+=`` (conv dgrad's addend) do so."""

    def __init__(self, root, root_needs_grad):
        self.ops = []
        self.root = root
        self.root_needs_grad = root_needs_grad

    def record(self, inputs, output, bwd):
        self.ops.append((inputs, output, bwd))

    def needs_grad(self, t):
        return t is not self.root or self.root_needs_grad

    def backward(self, out, dout):
        grads = {id(out): dout}
        while self.ops:
            inputs, output, bwd = self.ops.pop()
            g = grads.pop(id(output), None)
            if g is None:
                continue
            existing = [None if t is None else grads.get(id(t)) for t in inputs]
            new = bwd(g, existing)
            for t, ng in zip(inputs, new):
                if t is not None and ng is not None:
                    grads[id(t)] = ng
        return grads.get(id(self.root))


def _accum(existing, fresh):
    """Generic fallback when a producer cannot fuse the accumulation."""
    if existing is None:
        return fresh
    return ops.add_(existing, fresh)


def grad_of(p):
    """The kernels accumulate into p.grad; create it zeroed on first use (layout = p's layout)."""
    if p.grad is None:
        p.grad = ops.fill_(torch.empty_like(p), 0.0)
    return p.grad


# ------------------------------------------------------------------------------------------- ops on the tape
def conv(tape, x, weight, stride, pad, bias=None):
    y = ops.conv2d_fwd(x, weight, stride, pad, bias=bias)
    if tape is not None:
        need_dx = tape.needs_grad(x)

        def bwd(dy, existing):
            ops.conv2d_wgrad(x, dy, weight, grad_of(weight), stride, pad, accumulate=True)
            if bias is not None:
                ops.colsum(dy, grad_of(bias), accumulate=True)
            if not need_dx:
                return (None,)
            ex = existing[0]
            dx = ops.conv2d_dgrad(dy, weight, x.shape, stride, pad, addend=ex, out=ex)
            return (dx,)
        tape.record((x,), y, bwd)
    return y


def batchnorm(tape, x, bn, relu=False, residual=None):
    y, mean, invstd = ops.bn_train_fwd(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                       relu=relu, residual=residual, eps=bn.eps, momentum=bn.momentum)
    if tape is not None:
        def bwd(dy, existing):
            dx, dres = ops.bn_train_bwd(dy, y, x, bn.weight, mean, invstd, relu, grad_of(bn.weight), grad_of(bn.bias),
                                        want_dres=residual is not None, accumulate=True)
            if residual is None:
                return (_accum(existing[0], dx), None)
            return (_accum(existing[0], dx), _accum(existing[1], dres))
        tape.record((x, residual), y, bwd)
    return y


def maxpool(tape, x):
    y, am = ops.maxpool_fwd(x)
    if tape is not None:
        tape.record((x,), y, lambda dy, ex: (_accum(ex[0], ops.maxpool_bwd(dy, am, x.shape)),))
    return y


def global_avgpool(tape, x):
    y = ops.gap_fwd(x)
    if tape is not None:
        tape.record((x,), y, lambda dy, ex: (_accum(ex[0], ops.gap_bwd(dy, x.shape)),))
    return y


def linear(tape, x, weight, bias):
    """x [B,Din] -> [B,Dout]: a 1x1 'convolution' over a 1x1 image on the same MFMA kernels."""
    b, din = x.shape
    x4 = x.view(b, 1, 1, din)
    y4 = ops.conv2d_fwd(x4, weight, 1, 0, bias=bias)
    y = y4.view(b, weight.shape[0])
    if tape is not None:
        need_dx = tape.needs_grad(x)

        def bwd(dy, existing):
            dy4 = dy.view(b, 1, 1, -1)
            ops.conv2d_wgrad(x4, dy4, weight, grad_of(weight), 1, 0, accumulate=True)
            if bias is not None:
                ops.colsum(dy, grad_of(bias), accumulate=True)
            if not need_dx:
                return (None,)
            ex = existing[0]
            ex4 = None if ex is None else ex.view(b, 1, 1, din)
            dx = ops.conv2d_dgrad(dy4, weight, x4.shape, 1, 0, addend=ex4, out=ex4)
            return (dx.view(b, din),)
        tape.record((x,), y, bwd)
    return y


def l2_normalize(tape, x):
    d = x.shape[1]
    zhat, inv = ops.l2norm_fwd(x, normalize=True)
    if tape is not None:
        tape.record((x,), zhat, lambda dy, ex: (_accum(ex[0], ops.l2norm_bwd(zhat, inv, dy.contiguous(), d, True)),))
    return zhat


# ------------------------------------------------------------------------------------------- autograd bridge
class _Bridge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, module, record):
        if not x.is_cuda:
            raise SsvError(f"{type(module).__name__}: the HIP path needs device tensors; there is no CPU fallback")
        xin = module._prepare_input(x.detach())
        tape = Tape(xin, x.requires_grad) if record else None
        y = module._run(tape, xin)
        ctx.tape, ctx.y, ctx.module = tape, y, module
        return y.view_as(y)        # fresh tensor object for autograd; same storage

    @staticmethod
    def backward(ctx, dy):
        tape = ctx.tape
        if tape is None:
            raise SsvError("backward through a forward that ran under torch.no_grad()")
        ctx.tape = None
        dx = tape.backward(ctx.y, dy.contiguous())
        if dx is not None:
            dx = ctx.module._finish_input_grad(dx)
        return None, dx, None, None


class HipModule(nn.Module):
    """Base of every module on the HIP path.  Sub-classes implement ``_run(tape, x)``."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_anchor", torch.zeros(1, requires_grad=True))

    def _prepare_input(self, x):
        return x.contiguous()

    def _finish_input_grad(self, dx):
        return dx

    def forward(self, x):
        # grad mode is read HERE: inside Function.forward autograd has already switched it off
        return _Bridge.apply(self._anchor, x, self, torch.is_grad_enabled())

    def train(self, mode=True):
        # the reference never leaves train mode; eval-mode BN is not part of the path
        if not mode:
            raise SsvError("eval() is not supported: the reference path always runs BatchNorm with batch statistics")
        return super().train(mode)


# ------------------------------------------------------------------------------------------- parameter holders
class HipConv2d(HipModule):
    """Bias-free convolution; ``weight`` is [O,I,k,k] in channels_last (OHWI) memory."""

    def __init__(self, cin, cout, k, stride=1, pad=0, weight=None):
        super().__init__()
        w = torch.empty(cout, cin, k, k) if weight is None else weight
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.stride, self.pad = stride, pad

    def _run(self, tape, x):
        return conv(tape, x, self.weight, self.stride, self.pad)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        if not self.weight.data.is_contiguous(memory_format=torch.channels_last):
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        return self


class HipBatchNorm(HipModule):
    """BatchNorm2d / BatchNorm1d parameters + buffers (train-mode statistics only)."""

    def __init__(self, c, eps=ops.BN_EPS, momentum=ops.BN_MOMENTUM):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))
        self.eps, self.momentum = eps, momentum

    def _run(self, tape, x):
        return batchnorm(tape, x, self)


class HipLinear(HipModule):
    def __init__(self, din, dout, weight=None, bias=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dout, din) if weight is None else weight)
        self.bias = nn.Parameter(torch.empty(dout) if bias is None else bias)

    def _run(self, tape, x):
        return linear(tape, x, self.weight, self.bias)
