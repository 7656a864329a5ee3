"""TFLOP/s of one conv shape vs batch (tile-count quantisation).  args: mode H C K R stride  B1 B2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssv_amd import ops
mode, H, C, K, R, s = sys.argv[1], *[int(v) for v in sys.argv[2:7]]
p = R // 2
dev = torch.device("cuda:0")
for B in [int(v) for v in sys.argv[7:]]:
    x = torch.randn(B, H, H, C, device=dev)
    w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd(x, w, s, p); dy = torch.randn_like(y); dw = torch.zeros_like(w)
    fn = {"fwd": lambda: ops.conv2d_fwd(x, w, s, p), "dgrad": lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p),
          "wgrad": lambda: ops.conv2d_wgrad(x, dy, w, dw, s, p, accumulate=True)}[mode]
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    M = y.numel() // K
    print(f"B={B:4d} M={M:7d} blocks128={((M+127)//128)*((K+127)//128):5d}  {t:7.3f} ms  {2.0*y.numel()*C*R*R/(t*1e-3)/1e12:6.1f} TF")
