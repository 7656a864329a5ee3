import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch
from ssv_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, rep=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
for (n, hw, c, k) in ((512, 56, 64, 256), (512, 28, 128, 512), (512, 14, 256, 1024), (512, 56, 256, 64)):
    x = torch.randn(n, hw, hw, c, device=dev)
    w = (torch.randn(k, 1, 1, c, device=dev) * 0.05).permute(0, 3, 1, 2)
    gb = (x.numel() + n * hw * hw * k) * 4 / 1e9
    t0 = timeit(lambda: ops.conv2d_fwd(x, w, 1, 0))
    t1 = timeit(lambda: ops.conv2d_fwd_stats(x, w, 1, 0))
    print(f"{hw}x{hw} {c}->{k}: plain {t0:.3f} ms ({gb / t0:.2f} TB/s)  stats {t1:.3f} ms ({gb / t1:.2f} TB/s)   HBM bound {gb / 6.29:.3f} ms", flush=True)
