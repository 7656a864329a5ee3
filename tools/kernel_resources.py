#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
    python tools/kernel_resources.py self-supervised-vision_amd/csrc/conv_mfma.hip [filter substring] [-- extra hipcc flags]
Columns: VGPRs (arch), AGPRs, SGPRs, scratch bytes/lane, waves/SIMD the compiler reports, LDS bytes/workgroup."""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
rest = sys.argv[2:]
extra = rest[rest.index("--") + 1:] if "--" in rest else []
flt = [a for a in (rest[:rest.index("--")] if "--" in rest else rest)]
with tempfile.TemporaryDirectory() as tmp:
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-function",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", f"{tmp}/o.o"] + extra
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip().split(" ")[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    n = re.sub(r"\(anonymous namespace\)::", "", d)
    n = re.sub(r"\((ConvKP|\(anonymous).*", "", n)
    n = re.sub(r"^void ", "", n)
    if flt and not all(f in n for f in flt):
        continue
    scr, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{n:100s} v{g('VGPRs'):4d} a{g('AGPRs'):4d} s{g('SGPRs'):4d} scr{scr:4d} occ{occ:2d} lds{lds:7d}")
