"""CPU oracle for the two-view self-supervised training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / reported baseline.
The product path (``self-supervised-vision_amd``) never imports this package
and fails loudly when its HIP library is missing.

The oracle is a plain PyTorch-fp32-on-CPU restatement of the reference
algorithm (NightShade99/Self-Supervised-Vision), written functionally over a
flat ``{state_dict key: tensor}`` parameter dictionary.  Every function cites
the reference file:line it restates.

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference
itself (``tests/golden/gen_golden.py``, run in the build container where
``/root/reference`` exists) and ``tests/test_oracle_golden.py`` checks this
restatement against them.  Exception: the augmentation chain (R1) lives in
torchvision==0.9.1 + Pillow==8.3.1, neither vendored nor installed - that
piece is "parity unpinned" against torchvision and is pinned against Pillow
(present here) per deterministic op instead; see ``oracle/augment.py``.
"""
from .nets import (  # noqa: F401
    RESNET_SPECS, init_resnet, resnet_forward, init_simclr_head, simclr_head_forward,
    init_byol_mlp, byol_mlp_forward, init_barlow_head, barlow_head_forward,
    init_linear, tensor_checksum,
)
from .losses import ntxent_loss, barlow_loss, byol_mse_loss, l2_normalize  # noqa: F401
from .optim import sgd_nesterov_step, seeded_lr, ema_update, byol_tau  # noqa: F401
from .step import SimCLROracle, BYOLOracle, BarlowOracle, twin64, snapshot  # noqa: F401
from . import evalknn  # noqa: F401,E402
