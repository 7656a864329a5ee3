#!/usr/bin/env python3
"""The 'where does the step's kernel time go' table from one per-variant counter table (tools/pmc_variants.py output) and the per-class HBM traffic of the same build
(tools/pmc_traffic.py output): the MFMA-bound GEMM variants (matrix pipe >= 0.6 busy), the HBM-bound GEMM variants (< 0.6: the formed-on-load / short-contraction 1x1
layers, the stride-2 shortcut data gradients), BatchNorm passes, Winograd transforms, the rest.
    python tools/where_time_goes.py profiles/r05_simclr_b512_kernel_variants.txt profiles/r05_simclr_b512_pmc_hbm_traffic.json"""
import json
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\S.*)$", line)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), m.group(8).strip()))
traffic = json.load(open(sys.argv[2]))["per_step_gb"] if len(sys.argv) > 2 else {}


def group(name, busy):
    if name.startswith(("conv_fwd_k", "conv_dgrad_k", "conv_wgrad_k", "stem_")):
        return "GEMM kernels, matrix pipe >= 0.6 busy" if busy >= 0.6 else "GEMM kernels, matrix pipe < 0.6 busy (HBM-bound variants)"
    if name.startswith(("wino", "wgrad_reduce")):
        return "Winograd transforms + split-K reduces"
    if name.startswith("bn_"):
        return "BatchNorm passes (element-wise, finalize, coarsen)"
    if name.startswith("attn_"):
        return "attention (fp32 MFMA, flash-style)"
    if name.startswith("ln_"):
        return "LayerNorm passes"
    return "rest (augmentation, pools, loss, optimizer, fills, filter transposes)"


agg = {}
for ms, n, ghz, busy, parked, name in rows:
    g = agg.setdefault(group(name, busy), [0.0, 0.0, 0.0, 0.0, 0.0])
    g[0] += ms; g[1] += n; g[2] += busy * ms; g[3] += ghz * ms; g[4] += parked * ms
total = sum(v[0] for v in agg.values())
print(f"(the table's input lists the {len(rows)} largest kernel variants)")
print(f"{'ms/step':>8s} {'share':>6s} {'launches':>8s} {'busy':>5s} {'GHz':>5s} {'parked':>6s}  group   (single stream, one counter pass: {total:.1f} ms of kernels per step)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{v[0]:8.1f} {v[0] / total:6.2f} {v[1]:8.0f} {v[2] / v[0]:5.2f} {v[3] / v[0]:5.2f} {v[4] / v[0]:6.2f}  {k}")
if traffic:
    print("HBM traffic of the same build per class (GB per step, fetch + write): " + ", ".join(f"{k} {v['fetch'] + v['write']:.1f}" for k, v in sorted(traffic.items(), key=lambda kv: -(kv[1]['fetch'] + kv[1]['write'])) if v['fetch'] + v['write'] >= 0.5)
          + f"; total {sum(v['fetch'] + v['write'] for v in traffic.values()):.1f}")
