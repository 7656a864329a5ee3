"""Run one conv shape repeatedly (for rocprofv3 --pmc).  args: mode(fwd|dgrad|wgrad) B H C K R stride reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssv_amd import ops
mode, B, H, C, K, R, s, reps = sys.argv[1], *[int(v) for v in sys.argv[2:9]]
p = R // 2
dev = torch.device("cuda:0")
x = torch.randn(B, H, H, C, device=dev)
w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y = ops.conv2d_fwd(x, w, s, p)
dy = torch.randn_like(y)
dw = torch.zeros_like(w)
for _ in range(reps):
    if mode == "fwd":
        ops.conv2d_fwd(x, w, s, p)
    elif mode == "dgrad":
        ops.conv2d_dgrad(dy, w, x.shape, s, p)
    else:
        ops.conv2d_wgrad(x, dy, w, dw, s, p, accumulate=True)
torch.cuda.synchronize()
