import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class _G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    return _G()


def seeded_randn(seed, *shape):
    import torch
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def corr_views(seed, b, size=224):
    """Two noisy views of one smooth random image per sample: inputs whose samples differ at low spatial frequencies, like real images.
    (i.i.d. noise images are a degenerate input for a BatchNorm network at 224 x 224: after global pooling all samples look alike.)"""
    import torch
    base = torch.nn.functional.interpolate(seeded_randn(seed, b, 3, 7, 7), size=size, mode="bilinear", align_corners=False) * 2.0
    return base + 0.3 * seeded_randn(seed + 1, b, 3, size, size), base + 0.3 * seeded_randn(seed + 2, b, 3, size, size)
