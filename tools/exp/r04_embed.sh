#!/bin/bash
# token assembly kernels at the DINO shapes: time per launch (events), shipped library
python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from ssv_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for b, size in ((512, 224), (2048, 96)):
    patch, e = 16, 192
    n = (size // patch) ** 2; p3 = 768
    img = torch.randn(b, size, size, 3, device=dev); cls = torch.randn(1, p3, device=dev); pos = torch.randn(n + 1, e, device=dev)
    tok, t = ops.vit_embed_fwd(img, cls, pos, patch)
    dtok = torch.randn_like(tok); dcls, dpos = torch.zeros_like(cls), torch.zeros_like(pos)
    tf = timed(lambda: ops.vit_embed_fwd(img, cls, pos, patch))
    tb = timed(lambda: ops.vit_embed_bwd(dtok, b, t, p3, e, dcls, dpos, accumulate=True))
    print("B %d %dx%d: token assembly forward %.0f us (%.2f TB/s written), backward %.0f us" % (b, size, size, tf, tok.numel() * 4 / tf / 1e6, tb))
PY
