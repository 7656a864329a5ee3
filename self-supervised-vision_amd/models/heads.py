"""Projector / predictor heads of the three accelerated algorithms, composed from the HIP ops.
Parameter names, shapes and init draws follow the reference so state_dicts and seeds interchange:
  SimCLR ProjectionHead  models/simclr.py:23-36   fc1 bn1 fc2 bn2
  BYOL   MLP             models/byol.py:24-34     fc1 bn1 fc2
  Barlow ProjectionHead  models/barlow.py:23-36   layer1.{0,1} layer2.{0,1} layer3, then L2-normalise
"""
import math

import torch
import torch.nn as nn

from .. import nn as hnn


def _fresh_linear(din, dout):
    """nn.Linear's default init: kaiming_uniform_(weight, a=sqrt 5) then bias ~ U(-1/sqrt(fan_in), +)."""
    w = torch.empty(dout, din)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(din)
    b = torch.empty(dout)
    nn.init.uniform_(b, -bound, bound)
    return hnn.HipLinear(din, dout, weight=w, bias=b)


class SimclrProjectionHead(hnn.HipModule):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.fc1 = _fresh_linear(input_dim, input_dim)
        self.bn1 = hnn.HipBatchNorm(input_dim)
        self.fc2 = _fresh_linear(input_dim, output_dim)
        self.bn2 = hnn.HipBatchNorm(output_dim)

    def _run(self, tape, x):
        x = hnn.batchnorm(tape, self.fc1._run(tape, x), self.bn1, relu=True)
        return hnn.batchnorm(tape, self.fc2._run(tape, x), self.bn2)


class ByolMLP(hnn.HipModule):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.fc1 = _fresh_linear(input_dim, input_dim)
        self.bn1 = hnn.HipBatchNorm(input_dim)
        self.fc2 = _fresh_linear(input_dim, output_dim)

    def _run(self, tape, x):
        return self.fc2._run(tape, hnn.batchnorm(tape, self.fc1._run(tape, x), self.bn1, relu=True))


class _LinearBnRelu(nn.Sequential):
    def __init__(self, din, dout):
        super().__init__(_fresh_linear(din, dout), hnn.HipBatchNorm(dout))     # the reference's third entry (ReLU) has no state

    def _run(self, tape, x):
        return hnn.batchnorm(tape, self[0]._run(tape, x), self[1], relu=True)


class BarlowProjectionHead(hnn.HipModule):
    def __init__(self, input_dim, projection_dim):
        super().__init__()
        self.layer1 = _LinearBnRelu(input_dim, projection_dim)
        self.layer2 = _LinearBnRelu(projection_dim, projection_dim)
        self.layer3 = _fresh_linear(projection_dim, projection_dim)

    def _run(self, tape, x):
        x = self.layer2._run(tape, self.layer1._run(tape, x))
        return hnn.l2_normalize(tape, self.layer3._run(tape, x))
