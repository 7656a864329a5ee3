#!/usr/bin/env python3
"""Timeline summary of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: per queue busy time, how long 0 / 1 / 2+ kernels
were in flight, and per kernel class (conv / element-wise BatchNorm / other) the time spent alone and beside a kernel of another queue.
    python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction=0.5]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ks = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), name))
ks.sort()
t0, t1 = ks[0][0], max(k[1] for k in ks)
cut = t0 + (t1 - t0) * skip                      # steady state only: the later part of the run
ks = [k for k in ks if k[0] >= cut]


def cls(n):
    if "conv_" in n and "wgrad_reduce" not in n:
        return "conv"
    if re.search(r"bn_apply_k|bn_bwd_apply_k", n):
        return "bn_apply"
    if "bn_" in n or "pool" in n or "aug_" in n:
        return "stream_other"
    return "other"


ev = []
for i, (s, e, q, n) in enumerate(ks):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
active = set()
last = ev[0][0]
depth_time = defaultdict(int)
alone = defaultdict(int)
alone_by_name = defaultdict(int)
beside = defaultdict(lambda: defaultdict(int))
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        qs = {ks[j][2] for j in active}
        depth_time[min(len(qs), 2)] += dt
        for j in active:
            c = cls(ks[j][3])
            others = {cls(ks[o][3]) for o in active if ks[o][2] != ks[j][2]}
            if not others:
                alone[c] += dt
                alone_by_name[re.sub(r"[<(].*", "", ks[j][3].replace("void ", "").replace("(anonymous namespace)::", ""))[:60]] += dt
            else:
                for oc in others:
                    beside[c][oc] += dt
    last = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
span = (max(k[1] for k in ks) - ks[0][0]) / 1e6
print(f"window {span:.1f} ms, {len(ks)} kernels")
print("queues in flight: " + ", ".join(f"{d}: {v / 1e6:.1f} ms" for d, v in sorted(depth_time.items())))
dur = defaultdict(lambda: [0, 0])
for s, e, q, n in ks:
    c = cls(n)
    dur[c][0] += e - s
    dur[c][1] += 1
for c, (tot, n) in sorted(dur.items()):
    print(f"{c:13} {n:6} launches, {tot / 1e6:8.1f} ms kernel time, avg {tot / n / 1e3:7.1f} us | alone {alone[c] / 1e6:7.1f} ms | beside: " +
          ", ".join(f"{oc} {v / 1e6:.1f}" for oc, v in sorted(beside[c].items())))
print("kernels that ran with no kernel of another queue beside them (top 15):")
for n, v in sorted(alone_by_name.items(), key=lambda kv: -kv[1])[:15]:
    print(f"  {v / 1e6:8.1f} ms  {n}")
