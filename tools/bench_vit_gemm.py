#!/usr/bin/env python3
"""The Linear-layer GEMMs of the DINO ViT-S/16 step alone (fp32 MFMA): forward, data gradient (forward kernel on the transposed weights) and weight
gradient (+ bias) for every (tokens, in, out) the step launches, ms and TFLOP/s.
    python tools/bench_vit_gemm.py [repeats = 10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssv_amd import ops  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for m in (100864, 75776):              # 512 global crops x 197 tokens, 2048 local crops x 37 tokens
    for cin, cout, name in ((384, 1152, "qkv"), (384, 1536, "fc1"), (1536, 384, "fc2")):
        x = torch.randn(m, 1, 1, cin, device=dev)
        w = (torch.randn(cout, 1, 1, cin, device=dev) * 0.05).permute(0, 3, 1, 2)
        b = torch.randn(cout, device=dev)
        dy = torch.randn(m, 1, 1, cout, device=dev)
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        gf = 2.0 * m * cin * cout / 1e9
        t_f = timeit(lambda: ops.conv2d_fwd(x, w, 1, 0, bias=b))
        t_d = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 0))
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, 1, 0, accumulate=True, dbias=db))
        tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
        print(f"tokens {m:6d} {name} {cin:4d}->{cout:4d} ({gf:6.1f} GFLOP): forward {t_f:6.3f} ms {gf / t_f:6.1f} TF | data gradient {t_d:6.3f} ms {gf / t_d:6.1f} TF"
              f" | weight+bias gradient {t_w:6.3f} ms {gf / t_w:6.1f} TF", flush=True)
print({k: round(v, 3) for k, v in tot.items()})
