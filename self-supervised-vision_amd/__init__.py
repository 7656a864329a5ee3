"""self-supervised-vision_amd: the two-view self-supervised training step of
NightShade99/Self-Supervised-Vision on AMD MI355X (gfx950) - hand-written HIP kernels behind a
C ABI (csrc/, include/ssv_hip.h) and a Python host that mirrors the reference's surface
(main.py flags, models/*.train_step, networks/resnet, utils/losses, utils/train_utils).

Import as ``ssv_amd``.  There is no CPU fallback: the library raises if libssv_hip.so is missing.
"""
__version__ = "0.1.0"
