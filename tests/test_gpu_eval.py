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


def test_linear_probe_matches_oracle(dev):
    """The GPU probe (MFMA logits / weight gradient, softmax-CE kernel, plain-momentum SGD kernel) against the torch CPU restatement
    with the same init draw and the same batch order."""
    from ssv_amd.utils import eval_utils
    xtr, ytr = evalknn.clustered_features(81, 700, 126, 10, 2.0)             # 126: not a multiple of 4 -> padded; 700: ragged last batch
    xte, yte = evalknn.clustered_features(81, 300, 126, 10, 2.0)
    cfg = {"epochs": 6, "batch_size": 128, "lr": 0.1, "input_dim": 128}
    torch.manual_seed(5)
    want, head = evalknn.linear_evaluation(cfg, {"fvecs": xtr, "labels": ytr}, {"fvecs": xte, "labels": yte}, 10, eval_utils.probe_batches)
    torch.manual_seed(5)
    got = eval_utils.linear_evaluation(cfg, {"fvecs": xtr, "labels": ytr}, {"fvecs": xte, "labels": yte}, 10, dev)
    assert abs(got - want) <= 1.0 / 300 + 1e-9, (got, want)
    assert want > 0.5                                                         # the probe actually learns the planted classes


def test_softmax_ce_kernel(dev):
    from ssv_amd import _lib
    n, c, ld = 37, 10, 12
    logits = torch.randn(n, ld, generator=torch.Generator().manual_seed(3)) * 3
    labels = torch.randint(0, c, (n,), generator=torch.Generator().manual_seed(4), dtype=torch.int32)
    ref_in = logits[:, :c].double().requires_grad_()
    ref = torch.nn.functional.nll_loss(torch.log_softmax(ref_in, -1), labels.long())
    ref.backward()
    lg, lb = logits.to(dev), labels.to(dev)
    stats, dl = torch.zeros(2, device=dev), torch.full((n, ld), 7.0, device=dev)
    ws = torch.empty(_lib.load().ssv_softmax_ce_workspace_bytes(n), dtype=torch.uint8, device=dev)
    _lib.call("ssv_softmax_ce_fwd_bwd", n, c, ld, _lib.ptr(lg), _lib.ptr(lb), _lib.ptr(stats), _lib.ptr(dl), _lib.ptr(ws), ws.numel(), _lib.stream())
    np.testing.assert_allclose(stats[0].item(), ref.item(), rtol=1e-6)
    np.testing.assert_allclose(stats[1].item(), float((ref_in.argmax(-1) == labels.long()).float().mean()), rtol=1e-6)
    np.testing.assert_allclose(dl[:, :c].cpu().numpy(), ref_in.grad.numpy(), rtol=1e-5, atol=1e-8)
    assert float(dl[:, c:].abs().max()) == 0.0
