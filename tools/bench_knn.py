#!/usr/bin/env python3
"""Time ssv_knn_label_agreement at CIFAR sizes (test 10k, train 50k; k=20) and report achieved GFLOP/s and S traffic.
    python tools/bench_knn.py [d,d,... = 128]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssv_amd import ops

dev = torch.device("cuda:0")
out = []
DIMS = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (128,)
for n, d in [(n, d) for d in DIMS for n in (10000, 50000)]:
    g = torch.Generator().manual_seed(n)
    z = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).to(dev)
    labels = torch.randint(0, 10, (n,), generator=g, dtype=torch.int32).to(dev)
    ops.knn_label_agreement(z, labels, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        c = ops.knn_label_agreement(z, labels, 20)
    dt = (time.perf_counter() - t0) / reps
    out.append({"n": n, "d": d, "k": 20, "ms": round(dt * 1e3, 3), "gram_tflops": round(2.0 * n * n * d / dt / 1e12, 2),
                "s_matrix_gbs": round(2.0 * n * n * 4 / dt / 1e9, 1), "count": c})
print(json.dumps(out))
