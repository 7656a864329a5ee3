#!/usr/bin/env python3
"""One 1x1 convolution 64 -> 256 at 56x56 (bs 512), the reverse 256 -> 64, and a fill of the same output, five launches each: a target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssv_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(512, 56, 56, 64, device=dev)
w = (torch.randn(256, 1, 1, 64, device=dev) * 0.05).permute(0, 3, 1, 2)
x2 = torch.randn(512, 56, 56, 256, device=dev)
w2 = (torch.randn(64, 1, 1, 256, device=dev) * 0.05).permute(0, 3, 1, 2)
for _ in range(5):
    y = ops.conv2d_fwd(x, w, 1, 0)
    ops.fill_(y, 1.0)
    y2 = ops.conv2d_fwd(x2, w2, 1, 0)
torch.cuda.synchronize()
