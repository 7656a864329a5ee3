"""GPU: kNN evaluation kernel (ssv_knn_label_agreement) against the reference's compute_neighbor_accuracy goldens and the oracle."""
import numpy as np
import pytest
import torch

from oracle import evalknn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_knn_accuracy_matches_reference_goldens(dev, golden):
    from ssv_amd.utils import eval_utils
    g = golden["eval_level"]
    for (seed, n, d, classes, spread, k), want in zip(g["knn_cases"], g["knn_accuracy"]):
        fvecs, labels = evalknn.clustered_features(int(seed), int(n), int(d), int(classes), float(spread))
        got = eval_utils.compute_neighbor_accuracy(fvecs, labels, k=int(k))
        # the n x k table is integer; a rounding-level near-tie may swap one boundary neighbour
        assert abs(got - want) <= 2.0 / (n * k), (seed, got, want)


def test_knn_counts_bit_exact_on_exactly_representable_scores(dev):
    """Small-integer features: every inner product is exact in fp32, so ordering (with ties by index) is fully determined."""
    from ssv_amd import ops
    g = torch.Generator().manual_seed(5)
    for n, d, k in ((300, 16, 20), (1000, 8, 7), (65, 4, 40), (4100, 12, 20), (21, 4, 20)):
        z = torch.randint(-3, 4, (n, d), generator=g).float()
        labels = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32)
        want = evalknn.neighbor_agreement_count(z.numpy(), labels.numpy(), k)
        assert ops.knn_label_agreement(z.to(dev), labels.to(dev), k) == want, (n, d, k)


def test_knn_large_set_multi_chunk_property(dev):
    """n above one S chunk (4096 rows): planted duplicates - every vector appears exactly twice with the same label and is far
    from everything else, so with k=1 (after dropping the best hit) the agreement count is exactly n."""
    from ssv_amd import ops
    n_half, d = 6000, 64
    g = torch.Generator().manual_seed(9)
    base = torch.nn.functional.normalize(torch.randn(n_half, d, generator=g), dim=1)
    z = torch.cat([base, base])
    labels = torch.arange(n_half, dtype=torch.int32).repeat(2)
    assert ops.knn_label_agreement(z.to(dev), labels.to(dev), 1) == 2 * n_half


def test_knn_rejects_bad_arguments(dev):
    from ssv_amd import _lib, ops
    z = torch.randn(10, 8, device=dev)
    labels = torch.zeros(10, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.SsvError):
        ops.knn_label_agreement(z, labels, 10)          # k must be < n
    with pytest.raises(_lib.SsvError):
        ops.knn_label_agreement(z.cpu(), labels.cpu(), 3)
