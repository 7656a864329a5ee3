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


def _fresh_conv(cin, cout, k, stride, pad):
    """nn.Conv2d's construction-time draw (kaiming_uniform_, a=sqrt 5) - it is overwritten later by
    kaiming_normal_, but it must be drawn to keep the global RNG stream aligned with the reference."""
    w = torch.empty(cout, cin, k, k)
    nn.init.kaiming_uniform_(w, a=_KAIMING_A)
    return hnn.HipConv2d(cin, cout, k, stride, pad, weight=w)


class _Downsample(nn.Sequential):
    """conv1x1(stride) -> BN, registered as '0' and '1' like the reference nn.Sequential."""

    def __init__(self, cin, cout, stride):
        super().__init__(_fresh_conv(cin, cout, 1, stride, 0), hnn.HipBatchNorm(cout))


class BasicBlock(hnn.HipModule):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, downsample=None, base_width=64):
        super().__init__()
        if base_width != 64:
            raise ValueError("BasicBlock only supports base_width = 64")
        self.conv1 = _fresh_conv(in_planes, planes, 3, stride, 1)
        self.bn1 = hnn.HipBatchNorm(planes)
        self.conv2 = _fresh_conv(planes, planes, 3, 1, 1)
        self.bn2 = hnn.HipBatchNorm(planes)
        self.downsample = downsample

    def _run(self, tape, x):
        out = hnn.batchnorm(tape, self.conv1._run(tape, x), self.bn1, relu=True)
        out = self.conv2._run(tape, out)
        identity = x
        if self.downsample is not None:
            identity = hnn.batchnorm(tape, self.downsample[0]._run(tape, x), self.downsample[1])
        return hnn.batchnorm(tape, out, self.bn2, relu=True, residual=identity)


class Bottleneck(hnn.HipModule):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, downsample=None, base_width=64):
        super().__init__()
        width = int(planes * base_width / 64)
        self.conv1 = _fresh_conv(in_planes, width, 1, 1, 0)
        self.bn1 = hnn.HipBatchNorm(width)
        self.conv2 = _fresh_conv(width, width, 3, stride, 1)       # stride sits on the 3x3
        self.bn2 = hnn.HipBatchNorm(width)
        self.conv3 = _fresh_conv(width, planes * 4, 1, 1, 0)
        self.bn3 = hnn.HipBatchNorm(planes * 4)
        self.downsample = downsample

    def _run(self, tape, x):
        out = hnn.batchnorm(tape, self.conv1._run(tape, x), self.bn1, relu=True)
        out = hnn.batchnorm(tape, self.conv2._run(tape, out), self.bn2, relu=True)
        out = self.conv3._run(tape, out)
        identity = x
        if self.downsample is not None:
            identity = hnn.batchnorm(tape, self.downsample[0]._run(tape, x), self.downsample[1])
        return hnn.batchnorm(tape, out, self.bn3, relu=True, residual=identity)


class ResNet(hnn.HipModule):
    """forward(img [B,3,H,W] fp32, NCHW or channels_last) -> features [B, 512*expansion]."""

    def __init__(self, block, layers, num_classes=10, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, reduce_bottom_conv=False):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped convolutions (resnext) are outside the accelerated path")
        if replace_stride_with_dilation not in (None, [False, False, False], (False, False, False)):
            raise NotImplementedError("dilated ResNets are outside the accelerated path")
        if norm_layer is not None:
            raise NotImplementedError("custom norm layers are outside the accelerated path")
        self.in_planes, self.base_width = 64, width_per_group
        self.conv1 = _fresh_conv(3, 64, 3, 1, 1) if reduce_bottom_conv else _fresh_conv(3, 64, 7, 2, 3)
        self.bn1 = hnn.HipBatchNorm(64)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.out_dim = 512 * block.expansion
        # second pass of the reference ctor: every conv re-drawn with kaiming_normal_(fan_out, relu)
        for m in self.modules():
            if isinstance(m, hnn.HipConv2d):
                w = torch.empty(m.weight.shape)              # draw on a contiguous tensor, like nn.Conv2d.weight
                nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                m.weight.data = w.contiguous(memory_format=torch.channels_last)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.in_planes != planes * block.expansion:
            downsample = _Downsample(self.in_planes, planes * block.expansion, stride)   # drawn BEFORE the block's convs
        layers = [block(self.in_planes, planes, stride, downsample, self.base_width)]
        self.in_planes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, planes, base_width=self.base_width))
        return nn.Sequential(*layers)

    def _prepare_input(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected an image batch [B,3,H,W], got {tuple(x.shape)}")
        from .. import ops
        return ops.nchw_to_nhwc(x)

    def _run(self, tape, x):
        x = hnn.batchnorm(tape, self.conv1._run(tape, x), self.bn1, relu=True)
        x = hnn.maxpool(tape, x)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                x = blk._run(tape, x)
        return hnn.global_avgpool(tape, x)


def resnet18(**kwargs):
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(**kwargs):
    return ResNet(BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50(**kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(**kwargs):
    return ResNet(Bottleneck, [3, 4, 23, 3], **kwargs)


def resnet152(**kwargs):
    return ResNet(Bottleneck, [3, 8, 36, 3], **kwargs)


def resnext50_32x4d(**kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], groups=32, width_per_group=4, **kwargs)


def resnext101_32x8d(**kwargs):
    return ResNet(Bottleneck, [3, 4, 23, 3], groups=32, width_per_group=8, **kwargs)


def wide_resnet50_2(**kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], width_per_group=128, **kwargs)


def wide_resnet101_2(**kwargs):
    return ResNet(Bottleneck, [3, 4, 23, 3], width_per_group=128, **kwargs)
