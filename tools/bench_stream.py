#!/usr/bin/env python3
"""Ceiling of the streamed data path (utils/data_utils.py): samples/s the host gather + H2D upload of uint8 images sustains, and the
samples/s of the loader as a whole (gather + upload + both augmented views on the GPU) with nothing else on the device.
    python tools/bench_stream.py [images = 20000] [size = 224] [batch = 512] [chunk batches = 8]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssv_amd.utils import data_utils  # noqa: E402

n, size, batch, kb = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 20000), (2, 224), (3, 512), (4, 8)))
dev = torch.device("cuda:0")
imgs = np.random.default_rng(0).integers(0, 256, size=(n, size, size, 3), dtype=np.uint8)
labels = np.zeros(n, dtype=np.int64)
norm = {"mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}
tfs = {"train": {"color_jitter": {"brightness": 0.4, "contrast": 0.4, "saturation": 0.4, "hue": 0.1, "apply_prob": 0.8}, "random_gray": {"p": 0.2},
                 "random_resized_crop": {"size": [size, size], "scale": [0.2, 1.0]}, "random_flip": None, "to_tensor": None, "normalize": norm},
       "test": {"center_crop": {"size": [size, size]}, "to_tensor": None, "normalize": norm}}
ld = data_utils.GpuTwoViewLoader(imgs, labels, tfs, batch_size=batch, shuffle=True, device=dev, max_resident_bytes=1, stream_chunk_batches=kb)
assert ld.streamed
print(f"host threads {torch.get_num_threads()} of {os.cpu_count()} cpus")
# raw gather + upload of one chunk
ids = torch.randperm(n)[:kb * batch]
stage = torch.empty((ids.numel(),) + imgs.shape[1:], dtype=torch.uint8).pin_memory()
devbuf = torch.empty_like(stage, device=dev)
for _ in range(3):
    t0 = time.perf_counter(); data_utils._gather_rows(ld.host, ids, stage); t1 = time.perf_counter()
    devbuf.copy_(stage, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
gb = stage.numel() / 1e9
print(f"chunk of {ids.numel()} images ({gb:.2f} GB): host gather {t1 - t0:.3f} s = {gb / (t1 - t0):.1f} GB/s, H2D {t2 - t1:.3f} s = {gb / (t2 - t1):.1f} GB/s "
      f"-> {ids.numel() / max(t1 - t0, t2 - t1):.0f} images/s when overlapped, {ids.numel() / (t2 - t0):.0f} serial")
for epoch in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); cnt = 0
    for b in ld:
        cnt += b["index"].numel()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"epoch {epoch}: {cnt} samples through the streamed loader (gather + upload + two augmented views + centre view) in {dt:.2f} s = {cnt / dt:.0f} samples/s")
