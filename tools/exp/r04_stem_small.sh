#!/bin/bash
# the stem kernels on the CIFAR-size stem (3x3 / 1, 32x32, bs 512) and a 96x96 crop stem (7x7 / 2): rows-in-LDS weight gradient (shipped) against the row-taps gather
for v in shipped nostemrows; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  python - <<PY
import torch, sys
sys.path.insert(0, ".")
from ssv_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for (n, h, r, s, p) in ((512, 32, 3, 1, 1), (512, 96, 7, 2, 3), (64, 224, 7, 2, 3)):
    x = torch.randn(n, h, h, 3, device=dev); w = (torch.randn(64, 3, r, r, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    wrows = ops.stem_weight_rows(w)
    y, _ = ops.stem_conv_fwd(x, wrows, tuple(w.shape), s, p, want_stats=True); dy = torch.randn_like(y)
    tf = timed(lambda: ops.stem_conv_fwd(x, wrows, tuple(w.shape), s, p, want_stats=True))
    tw = timed(lambda: ops.stem_conv_wgrad(x, dy, tuple(w.shape), s, p))
    print("$v  n %d %dx%d %dx%d/%d: forward %.0f us  weight gradient %.0f us" % (n, h, h, r, r, s, tf, tw))
PY
done
