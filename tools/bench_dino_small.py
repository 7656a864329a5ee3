#!/usr/bin/env python3
"""Step time of the reference's own DINO configuration (configs/dino.yaml: ViT hidden 384 / 6 layers / patch 4 on 32x32 global and 8x8 local crops, 2 + 6 views per copy,
bs 64) launched kernel by kernel and replayed as one HIP graph (ssv_amd.graph.StepGraph).      python tools/bench_dino_small.py [batch = 64]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ssv_amd.graph import StepGraph  # noqa: E402

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ENC = {"hidden_dim": 384, "embedding_dim": 192, "intermediate_dim": 768, "num_attention_heads": 6, "patch_size": 4,
       "num_local_patches": 4, "num_global_patches": 64, "num_encoder_layers": 6}                      # configs/dino.yaml
bench.BENCH_CFG["dino"] = dict(bench.BENCH_CFG["dino"], encoder=ENC, proj_head={"hidden_dim": 512, "proj_dim": 1024},
                               optimizer={"name": "adamw", "lr": 1e-4, "amsgrad": False, "epsilon": 1e-6, "weight_decay": 0.04})
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda v, sz: torch.randn(bs, v, 3, sz, sz, device=dev, generator=g)
batch = {"global_1": mk(2, 32), "global_2": mk(2, 32), "local_1": mk(6, 8), "local_2": mk(6, 8)}
out = {}
for mode in ("eager", "hip_graph"):
    step, _ = bench.build(dev, "dino")
    sg = StepGraph(step.trainer, mode="1" if mode == "hip_graph" else "0")
    for _ in range(5):
        sg(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        sg(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    out[mode] = {"ms_per_step": round(dt * 1e3, 2), "images_per_s": round(bs / dt, 1), "state": sg.describe()}
print(json.dumps({"workload": f"DINO, the reference's configs/dino.yaml encoder (ViT 384 x 6 layers, patch 4), 2 copies x (2 global 32x32 + 6 local 8x8), bs {bs}, loss read every step", **out}))
