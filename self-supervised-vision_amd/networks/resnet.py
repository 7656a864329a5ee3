"""ResNet encoders on the HIP path (no fc head; returns pooled features).

Same constructor names, keyword arguments, parameter init (RNG stream) and state_dict keys as the
reference encoder file (networks/resnet.py:78-193) so checkpoints and seeds interchange; the
computation itself runs on libssv_hip's fp32-MFMA implicit-GEMM convolutions and fused
BatchNorm(+ReLU)(+residual) kernels in NHWC.
"""
import math

import torch
import torch.nn as nn

from .. import nn as hnn

_KAIMING_A = math.sqrt(5)


def _fresh_conv(cin, cout, k, stride, pad, groups=1):
    """nn.Conv2d's construction-time draw (kaiming_uniform_, a=sqrt 5) - it is overwritten later by
    kaiming_normal_, but it must be drawn to keep the global RNG stream aligned with the reference."""
    w = torch.empty(cout, cin // groups, k, k)
    nn.init.kaiming_uniform_(w, a=_KAIMING_A)
    return hnn.HipConv2d(cin, cout, k, stride, pad, weight=w, groups=groups)


class _Downsample(nn.Sequential):
    """conv1x1(stride) -> BN, registered as '0' and '1' like the reference nn.Sequential."""

    def __init__(self, cin, cout, stride):
        super().__init__(_fresh_conv(cin, cout, 1, stride, 0), hnn.HipBatchNorm(cout))


class _ResidualUnit(hnn.HipModule):
    """conv/BN chain described by `plan(in, planes, stride, base_width) -> [(cin, cout, k, stride, pad), ...]`, registered as
    conv1/bn1 .. convN/bnN; the last BN kernel also adds the shortcut and applies the closing ReLU."""
    expansion = 1

    @staticmethod
    def plan(cin, planes, stride, base_width, groups=1):
        raise NotImplementedError

    def __init__(self, in_planes, planes, stride=1, downsample=None, base_width=64, groups=1):
        super().__init__()
        specs = self.plan(in_planes, planes, stride, base_width, groups)
        self.depth = len(specs)
        for i, spec in enumerate(specs, start=1):
            setattr(self, f"conv{i}", _fresh_conv(*spec))
            setattr(self, f"bn{i}", hnn.HipBatchNorm(spec[1]))
        self.downsample = downsample

    def last_bn(self):
        return getattr(self, f"bn{self.depth}")

    def _run(self, tape, x):
        """conv -> BN -> ReLU -> conv chains never write the activation between the convolutions: the BatchNorm in the middle only
        finalises its statistics (hnn.batchnorm(lazy=True)) and the next convolution applies scale / shift / ReLU while it stages its
        input.  The projection shortcut's BatchNorm is folded into the unit's closing kernel the same way."""
        # The unit's input may be the previous unit's closing activation that nobody has written yet (hnn.LazySum): conv1 forms and writes it
        # when it is a 1x1 / stride-1 convolution, so it runs BEFORE the projection shortcut (which then reads the tensor).
        h1 = self.conv1._run(tape, x, bn_stats=True)
        shortcut = x
        if self.downsample is not None:                      # conv1x1 -> BN: no ReLU, its only consumer is the closing BatchNorm below
            fold = getattr(self, f"conv{self.depth}").has_stats_epilogue()
            # a stride-2 shortcut next to a wide 1x1 / stride-1 conv1 (every bottleneck stage entry): its data gradient stays compact and conv1's
            # data gradient - the last contribution to the unit input's gradient - adds it in its epilogue (hnn.conv, ops.StridedGrad)
            c1 = self.conv1
            compact = (c1.weight.shape[2] == 1 and c1.stride == 1 and c1.groups == 1 and c1.weight.shape[1] >= 128 and c1.weight.shape[1] % 4 == 0
                       and c1.weight.shape[0] % 32 == 0)
            shortcut = hnn.batchnorm(tape, self.downsample[0]._run(tape, x, bn_stats=True, compact_dx=compact), self.downsample[1], lazy=fold)
        h = x
        for i in range(1, self.depth + 1):
            h = h1 if i == 1 else getattr(self, f"conv{i}")._run(tape, h, bn_stats=True)      # every conv here is followed by its BatchNorm
            closing = i == self.depth
            nxt = None if closing else getattr(self, f"conv{i + 1}")
            fuse = nxt is not None and nxt.can_fuse_input(h.shape)       # h = this conv's output = the next conv's input (NHWC)
            h = hnn.batchnorm(tape, h, getattr(self, f"bn{i}"), relu=True, residual=shortcut if closing else None, lazy=fuse, defer=closing)
        return h


class BasicBlock(_ResidualUnit):
    @staticmethod
    def plan(cin, planes, stride, base_width, groups=1):
        if base_width != 64 or groups != 1:
            raise ValueError("BasicBlock only supports groups = 1 and base_width = 64")
        return [(cin, planes, 3, stride, 1), (planes, planes, 3, 1, 1)]


class Bottleneck(_ResidualUnit):
    expansion = 4

    @staticmethod
    def plan(cin, planes, stride, base_width, groups=1):
        mid = int(planes * base_width / 64) * groups
        return [(cin, mid, 1, 1, 0), (mid, mid, 3, stride, 1, groups), (mid, 4 * planes, 1, 1, 0)]      # stride (and the groups) sit on the 3x3


_STAGE_PLANES = (64, 128, 256, 512)
_UNSUPPORTED = {"replace_stride_with_dilation": (None, [False] * 3, (False,) * 3), "norm_layer": (None,)}


class ResNet(hnn.HipModule):
    """forward(img [B,3,H,W] fp32, NCHW or channels_last) -> features [B, 512*expansion].  Keyword surface of the
    reference constructor (grouped 3x3 convolutions = the ResNeXt rows run as dense block-diagonal convolutions); dilated /
    custom-norm variants are refused rather than silently computed differently."""

    def __init__(self, block, layers, num_classes=10, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, reduce_bottom_conv=False):
        super().__init__()
        given = dict(replace_stride_with_dilation=replace_stride_with_dilation, norm_layer=norm_layer)
        for key, allowed in _UNSUPPORTED.items():
            if given[key] not in allowed:
                raise NotImplementedError(f"{key}={given[key]!r} is outside the accelerated path")
        stem = (3, 64, 3, 1, 1) if reduce_bottom_conv else (3, 64, 7, 2, 3)
        self.conv1, self.bn1 = _fresh_conv(*stem), hnn.HipBatchNorm(64)
        cin = 64
        for idx, (planes, depth) in enumerate(zip(_STAGE_PLANES, layers), start=1):
            stage, cin = self._stage(block, cin, planes, depth, 1 if idx == 1 else 2, width_per_group, groups)
            setattr(self, f"layer{idx}", stage)
        self.out_dim = cin
        self._reference_init(zero_init_residual)

    @staticmethod
    def _stage(block, cin, planes, depth, stride, base_width, groups=1):
        cout = planes * block.expansion
        # the projection shortcut is constructed (and its weights drawn) BEFORE the unit's own convs
        shortcut = _Downsample(cin, cout, stride) if (stride != 1 or cin != cout) else None
        units = [block(cin, planes, stride, shortcut, base_width, groups)]
        units += [block(cout, planes, base_width=base_width, groups=groups) for _ in range(depth - 1)]
        return nn.Sequential(*units), cout

    def _reference_init(self, zero_init_residual):
        """Second pass of the reference ctor: every conv re-drawn with kaiming_normal_(fan_out, relu) in modules() order."""
        for m in self.modules():
            if isinstance(m, hnn.HipConv2d):
                w = torch.empty(m.weight.shape)              # draw on a contiguous tensor, like nn.Conv2d.weight
                nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                m.weight.data = w.contiguous(memory_format=torch.channels_last)
            elif zero_init_residual and isinstance(m, _ResidualUnit):
                nn.init.zeros_(m.last_bn().weight)

    def _prepare_input(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected an image batch [B,3,H,W], got {tuple(x.shape)}")
        from .. import ops
        return ops.nchw_to_nhwc(x)

    def grad_stages(self):
        """Gradient buckets of the data-parallel exchange ("launched per bucket as wgrad finishes, layer4 first"): the stem and the
        four stages, contiguous parameter runs of the optimizer's arena."""
        return [list(self.conv1.parameters()) + list(self.bn1.parameters())] + \
               [list(getattr(self, f"layer{idx}").parameters()) for idx in range(1, len(_STAGE_PLANES) + 1)]

    def _run(self, tape, x):
        hnn.stage_mark(tape, self, 0)
        x = hnn.bn_relu_maxpool(tape, self.conv1._run(tape, x, bn_stats=True), self.bn1)      # one pass each way when the conv left statistics
        for idx in range(1, len(_STAGE_PLANES) + 1):
            hnn.stage_mark(tape, self, idx)      # fires in backward once this stage's weight gradients are enqueued
            for unit in getattr(self, f"layer{idx}"):
                x = unit._run(tape, x)
        return hnn.global_avgpool(tape, x)


# name -> (unit, units per stage, fixed constructor keywords)
_ZOO = {
    "resnet18": (BasicBlock, (2, 2, 2, 2), {}),
    "resnet34": (BasicBlock, (3, 4, 6, 3), {}),
    "resnet50": (Bottleneck, (3, 4, 6, 3), {}),
    "resnet101": (Bottleneck, (3, 4, 23, 3), {}),
    "resnet152": (Bottleneck, (3, 8, 36, 3), {}),
    "resnext50_32x4d": (Bottleneck, (3, 4, 6, 3), dict(groups=32, width_per_group=4)),
    "resnext101_32x8d": (Bottleneck, (3, 4, 23, 3), dict(groups=32, width_per_group=8)),
    "wide_resnet50_2": (Bottleneck, (3, 4, 6, 3), dict(width_per_group=128)),
    "wide_resnet101_2": (Bottleneck, (3, 4, 23, 3), dict(width_per_group=128)),
}


def _register(name):
    unit, depths, fixed = _ZOO[name]

    def build(**kwargs):
        return ResNet(unit, list(depths), **{**kwargs, **fixed})
    build.__name__ = build.__qualname__ = name
    build.__doc__ = f"{name} encoder (reference networks/resnet.py factory of the same name)."
    return build


for _name in _ZOO:
    globals()[_name] = _register(_name)
del _name
