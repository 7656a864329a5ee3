"""Diagnostic: per-phase cycle shares of the single-buffer conv k-loop (needs tools/probe/libssv_hip_stamp.so)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SSV_HIP_LIB"] = os.path.join(ROOT, "tools", "probe", "libssv_hip_stamp.so")
sys.path.insert(0, ROOT)
import torch
from ssv_amd import ops, _lib
lib = _lib.load()
mode, B, H, C, K, R, s = sys.argv[1], *[int(v) for v in sys.argv[2:8]]
p = R // 2
dev = torch.device("cuda:0")
x = torch.randn(B, H, H, C, device=dev)
w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y = ops.conv2d_fwd(x, w, s, p); dy = torch.randn_like(y); dw = torch.zeros_like(w)
fn = {"fwd": lambda: ops.conv2d_fwd(x, w, s, p), "dgrad": lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p),
      "wgrad": lambda: ops.conv2d_wgrad(x, dy, w, dw, s, p, accumulate=True)}[mode]
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.ssv_debug_stamps(buf, 1)
fn(); torch.cuda.synchronize()
lib.ssv_debug_stamps(buf, 1)
ld, mma, b1, st, b2, nk = [buf[i] for i in range(6)]
tot = ld + mma + b1 + st + b2
print(f"{mode} B={B} H={H} C={C} K={K} R={R} s={s}: k-tiles(sum over blocks)={nk}")
for name, v in (("issue loads", ld), ("frag reads + MFMA", mma), ("barrier 1", b1), ("vmcnt wait + ds_write", st), ("barrier 2", b2)):
    print(f"   {name:24s} {v / nk:9.1f} cycles/k-tile  {100.0 * v / tot:5.1f} %")
print(f"   total {tot / nk:9.1f} cycles/k-tile (ideal MFMA: {64 * 64 if True else 0} for BK=32 2x2)")
