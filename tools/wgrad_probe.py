import sys, os
sys.path.insert(0, "/root/repo")
import torch
from ssv_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, rep=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
B = 128
for (H, C, K) in ((28, 128, 128), (14, 256, 256), (56, 64, 64), (7, 512, 512)):
    # 3x3 stride 1 pad 1
    x = torch.randn(B, H, H, C, device=dev); dy = torch.randn(B, H, H, K, device=dev)
    w3 = torch.zeros(K, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last); dw3 = torch.zeros_like(w3)
    t3 = timeit(lambda: ops.conv2d_wgrad(x, dy, w3, dw3, 1, 1, accumulate=True))
    # same GEMM as a 1x1 over 9C channels
    x1 = torch.randn(B, H, H, 9 * C, device=dev)
    w1 = torch.zeros(K, 9 * C, 1, 1, device=dev).contiguous(memory_format=torch.channels_last); dw1 = torch.zeros_like(w1)
    t1 = timeit(lambda: ops.conv2d_wgrad(x1, dy, w1, dw1, 1, 0, accumulate=True))
    flop = 2.0 * B * H * H * K * C * 9
    print(f"H={H} C={C} K={K}: 3x3 wgrad {t3:.3f} ms {flop/t3/1e9:.1f} TF | 1x1 over 9C {t1:.3f} ms {flop/t1/1e9:.1f} TF")
