"""LayerNorm forward / backward alone at the DINO ViT-S/16 shapes (networks/vit.py:22-31,43-46: out = f(x) + LayerNorm(x)): time per launch and the HBM rate of
the every-operand-once byte count (forward: x, addend in, y out; backward: dy, x, addend in, dx out).  python tools/probe/ln_probe.py [C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ssv_amd import _lib, ops  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    dev = torch.device("cuda:0")
    print("library %s sources %s" % (_lib.lib_sha16(), _lib.source_sha16()))
    # bs 128, two augmented copies, 2 global + 8 local crops each (configs/dino_vits16_224_synthetic.yaml); operands of 155 / 116 MB each: the 256 MB
    # Infinity Cache cannot hold a launch's streams, as in the step
    for m, what in ((2 * 2 * 128 * 197, "global crops: 512 sequences x 197 tokens"), (2 * 8 * 128 * 37, "local crops: 2048 sequences x 37 tokens")):
        g = torch.Generator(device="cpu").manual_seed(m)
        x, add, dy = (torch.randn(m, c, generator=g).to(dev) for _ in range(3))
        gamma, beta = torch.rand(c, generator=g).to(dev) + 0.5, torch.randn(c, generator=g).to(dev)
        y, mean, invstd = ops.layernorm_fwd(x, gamma, beta, add)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        acc = add.clone()
        tf = timed(lambda: ops.layernorm_fwd(x, gamma, beta, add))
        tf0 = timed(lambda: ops.layernorm_fwd(x, gamma, beta, None))
        tb = timed(lambda: ops.layernorm_bwd(dy, x, gamma, mean, invstd, dg, db, dx_addend=acc, accumulate=True))
        tb0 = timed(lambda: ops.layernorm_bwd(dy, x, gamma, mean, invstd, dg, db, dx_addend=None, accumulate=False))
        byt = m * c * 4
        print("M %6d C %4d (%s): fwd+addend %6.1f us %5.2f TB/s | fwd %6.1f us %5.2f TB/s | bwd+addend %6.1f us %5.2f TB/s | bwd %6.1f us %5.2f TB/s" % (
            m, c, what, tf, 3 * byt / tf / 1e6, tf0, 2 * byt / tf0 / 1e6, tb, 4 * byt / tb / 1e6, tb0, 3 * byt / tb0 / 1e6))


if __name__ == "__main__":
    main()
