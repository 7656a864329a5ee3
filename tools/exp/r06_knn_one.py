"""one timed configuration of the fused kNN search (diagnostics): python tools/exp/r06_knn_one.py [n] [d]"""
import sys, time, torch
sys.path.insert(0, ".")
from ssv_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
k = 20
g = torch.Generator(device=dev).manual_seed(n)
z = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
labels = torch.randint(0, 10, (n,), device=dev, generator=g, dtype=torch.int32)
c = ops.knn_label_agreement(z, labels, k); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): c = ops.knn_label_agreement(z, labels, k)
print("  n %d d %d ms %.3f count %d" % (n, d, (time.perf_counter() - t0) / 5 * 1e3, c))
