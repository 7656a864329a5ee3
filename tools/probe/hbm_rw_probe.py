#!/usr/bin/env python3
"""HBM bandwidth by direction: write-only (fill), read-only (a reduction), read + write (copy) on 1.6 GB tensors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
n = 512 * 56 * 56 * 256
x = torch.randn(n, device=dev); y = torch.empty_like(x)
gb = n * 4 / 1e9
t = timeit(lambda: ops.fill_(y, 1.0)); print(f"write only  (ssv_fill)      {gb / t:6.2f} TB/s")
t = timeit(lambda: y.fill_(2.0)); print(f"write only  (torch fill_)   {gb / t:6.2f} TB/s")
t = timeit(lambda: y.copy_(x)); print(f"read + write (torch copy_)   {2 * gb / t:6.2f} TB/s total")
t = timeit(lambda: x.sum()); print(f"read only   (torch sum)      {gb / t:6.2f} TB/s")
cs = torch.zeros(256, device=dev)
t = timeit(lambda: ops.colsum(x.view(-1, 256), cs, accumulate=False)); print(f"read only   (ssv_colsum)     {gb / t:6.2f} TB/s")
